"""The checker: the GPU's first sweep and P against oracle/ (the only place, with the cpu_baseline leg, where bench.py
touches the oracle -- after the timed blocks)."""
from __future__ import annotations

import os

import numpy as np

from .common import PARITY_P_TOL, PARITY_TOL, WORKLOADS
from .ranks import Ranks


DEGENERATE_SOFTMAX_NOTE = (
    "reference-mode scores are dot / (||Z[src_all]||_F * ||Z[dst_all]||_F) (similarity.py:35-37: GLOBAL denominators): at "
    "|E| >= 4M they are O(1e-9), exp() of them is exactly 1 in fp32 and P = 1/deg on both sides, so parity_P_rel_l2_vs_oracle "
    "checks the softmax plumbing and the edge order, not K1's dot products.  Those are checked by "
    "parity_P_per_edge_rel_l2_vs_oracle here (the same kernels with per-edge cosine scores, all edges, against the C "
    "oracle) and in the GPU suite by test_k1_scores_at_scale / test_config3_per_edge_P_sampled_rows (raw dots, per-edge "
    "cosine, million-edge rows)")


def check_parity(args, ranks: Ranks, eng, m, csr, X, result, baselines: bool = True):
    """The first GPU sweep (and P itself) against the C oracle, which builds its OWN P; at N = 1 also the CPU
    baselines (the oracle timed on this box's host cores).  Returns (the oracle's first sweep on rank 0, failed) -- the
    verdict is all-reduced, so every rank leaves together."""
    dname = WORKLOADS[args.workload][4]
    Z1_oracle, failed = None, False
    P_gpu = eng.P_global() if eng.row_world == 1 else None      # row splits: each rank holds its rows of P
    if ranks.rank == 0:
        from oracle import baseline as B
        from oracle import clane_oracle as O
        from oracle import clane_oracle_c as OC
        if ranks.world > 1:     # torchrun gives every rank OMP_NUM_THREADS=1; the others are idle in the collective below
            OC.set_threads(os.cpu_count() or 1)
        Z1_oracle, first, _, P_oracle, Xf = B.oracle_first_sweep(csr, X, None, args.gamma)
        parity = O.rel_l2(m["Z1"].float(), Z1_oracle)
        result["parity_rel_l2_vs_oracle_after_1_sweep"] = parity
        result["parity_note"] = ("first sweep from Z = X on the GPU(s), P from the GPU build_P, against "
                                 "oracle/clane_oracle.c running its own build_P and sweep (fp32, on the bf16-rounded "
                                 "inputs where the storage is bf16); " + DEGENERATE_SOFTMAX_NOTE)
        failed = not parity < PARITY_TOL[dname]
        if P_gpu is not None:
            parity_p = O.rel_l2(P_gpu.float(), P_oracle)
            result["parity_P_rel_l2_vs_oracle"] = parity_p
            failed = failed or not parity_p < PARITY_P_TOL[dname]
        if ranks.world == 1 and not ranks.rehearsal and not args.no_cpu_baseline and not failed and baselines:
            result["cpu_baseline"] = B.cpu_baseline(csr, Xf, P_oracle, args.gamma, Z1_oracle, first)
            result["cpu_baseline_torch"] = B.cpu_baseline_torch(csr, Xf, P_oracle, args.gamma)
    return Z1_oracle, ranks.agree_to_fail(failed)


def per_edge_parity(args, ranks: Ranks, eng, csr, X, result) -> bool:
    """A K1 check that is NOT degenerate at full size (one GPU): the same engine scores every edge with the per-edge
    cosine (cosine_mode "per_edge": dot / (|z_src| |z_dst|), the cosine similarity.py's docstring describes) from Z = X
    and all of P is compared with the C oracle's per-edge build_P.  Leaves the engine with a reference-mode P of Z = X."""
    from oracle import clane_oracle as O
    from oracle import clane_oracle_c as OC
    dname = WORKLOADS[args.workload][4]
    eng.set_Z(X)
    eng.set_cosine_mode("per_edge")
    eng.build_P()
    P_pe = eng.P_global().float()
    eng.set_cosine_mode("reference")
    eng.build_P()
    P_or, _ = OC.build_P(csr.rowptr, csr.colidx, X.float(), mode="per_edge")
    err = O.rel_l2(P_pe, P_or)
    deg = np.diff(csr.rowptr)
    hub = int(np.argmax(deg))
    ph = P_or[int(csr.rowptr[hub]):int(csr.rowptr[hub + 1])]
    result["parity_P_per_edge_rel_l2_vs_oracle"] = err
    result["parity_P_per_edge_note"] = (f"all {csr.num_edges} values of P with per-edge cosine scores (not 1/deg: the "
                                        f"heaviest row's P spans a factor {float(ph.max() / ph.min()):.2f}) against "
                                        f"oracle/clane_oracle.c's per-edge build_P")
    return not err < PARITY_P_TOL[dname] * 5         # scores O(0.1): exp and the softmax sums see real arguments
