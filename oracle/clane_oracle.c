/*
 * clane_oracle.c -- plain-C restatement of CLANE's embedding path (TEST INFRASTRUCTURE ONLY).
 *
 * A second, independent oracle next to oracle/clane_oracle.py (PyTorch-CPU ops): explicit loops
 * over CSR, fixed summation order (edges of a row in column order, exactly the order of the
 * reference's per-vertex `weights.mm(z_nbrs)`), OpenMP over rows.  Used by tests/ (three-way check:
 * reference goldens vs Python oracle vs this file) and by bench.py's cpu_baseline leg.  Nothing
 * under clane_amd/ links or loads it.
 *
 * Reference citations (/root/reference):
 *   sweep        clane/embedder.py:84-94    z_v = x_v + gamma * sum_e P_e * Z_old[col_e]; rows without
 *                                           out-edges keep z (88-89); delta = sum |Z_new - Z_old| (94)
 *   edge scores  clane/graph.py:119-121 + clane/similarity.py:35-37   dot / (||Z[src_all]||_F * ||Z[dst_all]||_F)
 *   row softmax  clane/graph.py:122-123
 *
 * float32 storage and arithmetic (float accumulators, like torch's fp32 kernels up to order);
 * the two global norms and the delta are accumulated in double.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int clane_c_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Under torchrun every rank gets OMP_NUM_THREADS=1; the rank that runs the check may ask for the box's cores back. */
void clane_c_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* Z_new = X + gamma * P Z_old (rows without out-edges: Z_new = Z_old); returns sum |Z_new - Z_old|. */
double clane_c_sweep_f32(const int64_t *rowptr, const int32_t *colidx, const float *P, int64_t V, int32_t d,
                         const float *X, const float *Zold, float gamma, float *Znew) {
    double delta = 0.0;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : delta)
    for (int64_t v = 0; v < V; ++v) {
        const float *zo = Zold + v * d;
        float *zn = Znew + v * d;
        const int64_t e0 = rowptr[v], e1 = rowptr[v + 1];
        if (e0 == e1) {
            for (int32_t k = 0; k < d; ++k) zn[k] = zo[k];
            continue;
        }
        for (int32_t k = 0; k < d; ++k) zn[k] = 0.0f;
        for (int64_t e = e0; e < e1; ++e) {
            const float p = P[e];
            const float *zc = Zold + (int64_t)colidx[e] * d;
            for (int32_t k = 0; k < d; ++k) zn[k] += p * zc[k];
        }
        double row = 0.0;
        for (int32_t k = 0; k < d; ++k) {
            zn[k] = X[v * d + k] + gamma * zn[k];
            row += fabs((double)zn[k] - (double)zo[k]);
        }
        delta += row;
    }
    return delta;
}

/* P[e] = row-softmax of dot(z_src, z_dst) / D, D = sqrt(sum_e |z_src|^2) * sqrt(sum_e |z_dst|^2). Returns D.
 * per_edge != 0 (the build's extension, clane_amd cosine_mode="per_edge": the cosine CosineSimilarity's docstring
 * describes, similarity.py:9-19, instead of what line 37 computes): score = dot / (|z_src| * |z_dst|); returns 0.
 * Scores at the reference's global denominator are O(1e-9) on large graphs, where exp() is exactly 1 and P = 1/deg: this
 * mode is what makes a full-size comparison of P sensitive to the dot products. */
double clane_c_build_P_mode_f32(const int64_t *rowptr, const int32_t *colidx, int64_t V, int32_t d, const float *Z,
                                float *P, int per_edge) {
    double *sq = (double *)malloc(sizeof(double) * (size_t)V);
    int64_t *indeg = (int64_t *)calloc((size_t)V, sizeof(int64_t));
#pragma omp parallel for schedule(static)
    for (int64_t v = 0; v < V; ++v) {
        double s = 0.0;
        for (int32_t k = 0; k < d; ++k) s += (double)Z[v * d + k] * (double)Z[v * d + k];
        sq[v] = s;
    }
    for (int64_t e = 0; e < rowptr[V]; ++e) indeg[colidx[e]]++;
    double a = 0.0, b = 0.0;
    for (int64_t v = 0; v < V; ++v) {
        a += (double)(rowptr[v + 1] - rowptr[v]) * sq[v];
        b += (double)indeg[v] * sq[v];
    }
    const float D = sqrtf((float)a) * sqrtf((float)b); /* fp32 sqrt and product, like the reference */
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t v = 0; v < V; ++v) {
        const int64_t e0 = rowptr[v], e1 = rowptr[v + 1];
        if (e0 == e1) continue;
        const float *zs = Z + v * d;
        const float ns = sqrtf((float)sq[v]);
        float m = -INFINITY;
        for (int64_t e = e0; e < e1; ++e) {
            const float *zd = Z + (int64_t)colidx[e] * d;
            float dot = 0.0f;
            for (int32_t k = 0; k < d; ++k) dot += zs[k] * zd[k];
            P[e] = per_edge ? dot / (ns * sqrtf((float)sq[colidx[e]])) : dot / D;
            if (P[e] > m) m = P[e];
        }
        float s = 0.0f;
        for (int64_t e = e0; e < e1; ++e) {
            P[e] = expf(P[e] - m);
            s += P[e];
        }
        for (int64_t e = e0; e < e1; ++e) P[e] /= s;
    }
    free(sq);
    free(indeg);
    return per_edge ? 0.0 : (double)D;
}

double clane_c_build_P_f32(const int64_t *rowptr, const int32_t *colidx, int64_t V, int32_t d, const float *Z,
                           float *P) {
    return clane_c_build_P_mode_f32(rowptr, colidx, V, d, Z, P, 0);
}
