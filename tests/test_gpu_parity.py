"""GPU parity: the HIP path, called through the C ABI (clane_amd._hip.HipKernels), against the CPU
oracle on the same seeded inputs and against the committed golden vectors from the reference.

Tolerances (north_star: embeddings within 1e-4 relative L2 of the CPU reference):
  * fp32 stage outputs  rel-L2 <= 2e-6  (only summation order differs)
  * fp64 stage outputs  rel-L2 <= 1e-13
  * bf16 storage        rel-L2 <= 8e-3  (2^-8 rounding of the stored values)
  * final embeddings    rel-L2 <= 1e-5 fp32 (bar is 1e-4)
"""
import contextlib
import io
import os
from pathlib import Path

import numpy as np
import pytest
import torch

from clane_amd import _hip
from clane_amd.embedder import Embedder
from clane_amd.engine import SweepEngine
from clane_amd.graph import Graph
from clane_amd.partition import HostCSR
from clane_amd.similarity import CosineSimilarity
from clane_amd import synth
from oracle import clane_oracle as O
from oracle import clane_oracle_c as OC

from .conftest import load_golden, write_data_root
from .test_host_logic import KARATE_LIKE, graph_from_golden

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 2e-6, torch.float64: 1e-13, torch.bfloat16: 8e-3}


@pytest.fixture(scope="module")
def dev():
    return _hip.require_gpu("cuda:0")


@pytest.fixture(scope="module")
def k():
    return _hip.kernels()


def ragged_csr(V, seed, max_deg=40, hubs=(), empty_frac=0.2):
    """Random CSR with empty rows, ragged rows and a few very long rows (hubs: list of degrees)."""
    rng = np.random.default_rng(seed)
    deg = rng.integers(0, max_deg + 1, size=V)
    deg[rng.random(V) < empty_frac] = 0
    for i, h in enumerate(hubs):
        deg[(7 * i + 3) % V] = min(h, V)
    rowptr = np.zeros(V + 1, dtype=np.int64)
    np.cumsum(deg, out=rowptr[1:])
    colidx = np.concatenate([np.sort(rng.choice(V, size=dg, replace=False)) for dg in deg] + [np.empty(0, int)])
    return HostCSR(V, rowptr, colidx.astype(np.int32))


def padded(t, dtype, dev, pad_to=None):
    """[V, d] CPU -> [V, ld] device tensor with zero pad columns (the ABI's fast-path contract)."""
    V, d = t.shape
    ld = -(-d // _hip.VEC_ELEMS[dtype]) * _hip.VEC_ELEMS[dtype] if pad_to is None else pad_to
    out = torch.zeros(V, ld, dtype=dtype, device=dev)
    out[:, :d] = t.to(dtype)
    return out


def rel(a, b):
    return O.rel_l2(a.detach().cpu().double(), b.detach().cpu().double())


# ---- reference tests/test_similarity.py re-expressed on the HIP path ------------------------------
def test_cosine_known_answers(dev):
    cs = CosineSimilarity()
    g = load_golden("g1_cosine.npz")
    v = torch.tensor([1.0, 2.0, 3.0])
    assert cs(v, v).item() == pytest.approx(1.0, abs=1e-4)
    assert cs(torch.tensor([0.0, 1.0]), torch.tensor([1.0, 0.0])).item() == pytest.approx(0.0, abs=1e-4)
    assert cs(v, -v).item() == pytest.approx(-1.0, abs=1e-4)
    a, b = torch.from_numpy(g["a4"]), torch.from_numpy(g["b4"])
    out = cs(a, b)
    assert out.shape == torch.Size([4]) and out.device == a.device
    np.testing.assert_allclose(out.numpy(), g["out4"], rtol=1e-6)
    assert torch.allclose(cs(a[0], b[0]), cs(b[0], a[0]))
    out64 = cs(torch.from_numpy(g["a64"]), torch.from_numpy(g["b64"]))
    assert out64.dtype == torch.float64
    np.testing.assert_allclose(out64.numpy(), g["out64"], rtol=1e-14)
    assert torch.isnan(cs(torch.zeros(3), torch.ones(3))).all()             # D = 0 -> nan, like the reference
    big_a, big_b = torch.rand(5000, 1433), torch.rand(5000, 1433)
    np.testing.assert_allclose(cs(big_a, big_b).numpy(), O.cosine_similarity(big_a, big_b).numpy(), rtol=2e-5)
    assert cs(a.to(dev), b.to(dev)).is_cuda


# ---- stage kernels ---------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.bfloat16])
@pytest.mark.parametrize("d,pad", [(2, True), (16, True), (100, True), (128, True), (256, True), (1433, True),
                                   (1433, False), (3, False), (70, False)])
def test_row_sqnorm(dev, k, dtype, d, pad):
    V = 777
    Z = synth.gaussian_X(V, d, seed=d).to(dtype)
    Zd = padded(Z, dtype, dev) if pad else Z.to(dev).contiguous()
    sq = torch.empty(V, dtype=_hip.acc_dtype(dtype), device=dev)
    k.row_sqnorm(Zd, d, sq)
    ref = Z.double().pow(2).sum(1)                    # Z is already rounded to the storage dtype
    assert rel(sq, ref) < (1e-13 if dtype == torch.float64 else 1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.bfloat16])
@pytest.mark.parametrize("d,pad", [(2, True), (16, True), (128, True), (256, True), (300, True), (1433, True),
                                   (37, False)])
def test_edge_score_and_softmax_stages(dev, k, dtype, d, pad):
    csr = ragged_csr(500, seed=d, hubs=(64, 65, 130, 400))
    V = csr.num_vertices
    acc = _hip.acc_dtype(dtype)
    Zc = synth.gaussian_X(V, d, seed=3).to(dtype)
    Zf = Zc.to(acc)                                                    # what the kernel sees after widening
    Zd = padded(Zc, dtype, dev) if pad else Zc.to(dev).contiguous()
    rowptr, colidx = torch.from_numpy(csr.rowptr).to(dev), torch.from_numpy(csr.colidx).to(dev)
    scores = torch.full((csr.num_edges,), float("nan"), dtype=acc, device=dev)
    tol = 5e-6 if dtype == torch.bfloat16 else TOL[dtype]

    k.edge_score(rowptr, colidx, V, 0, Zd, d, _hip.SCORE_RAW_DOT, None, None, scores)
    dots = O.edge_dots(csr.rowptr, csr.colidx, Zf.double())
    assert rel(scores, dots) < tol
    # long rows cut into per-wave slices by the second launch: identical scores
    deg = np.diff(csr.rowptr)
    long_rows = torch.from_numpy(np.nonzero(deg > 48)[0].astype(np.int32)).to(dev)
    sliced = torch.full_like(scores, float("nan"))
    k.edge_score(rowptr, colidx, V, 0, Zd, d, _hip.SCORE_RAW_DOT, None, None, sliced, 48, long_rows)
    assert torch.equal(sliced, scores)

    # reference mode: denominators from K0 + degree-weighted reduction
    sq = torch.empty(V, dtype=acc, device=dev)
    k.row_sqnorm(Zd, d, sq)
    ws = torch.zeros(k.reduce_ws_len(), dtype=torch.float64, device=dev)
    sums2 = torch.zeros(2, dtype=torch.float64, device=dev)
    indeg = torch.from_numpy(csr.indeg()).to(dev)
    k.degree_weighted_sums(sq, rowptr, indeg, V, ws, sums2)
    D = O.global_denominator(csr.rowptr, csr.colidx, Zf.double())
    assert float(sums2[0].sqrt() * sums2[1].sqrt()) == pytest.approx(D, rel=1e-6)
    k.edge_score(rowptr, colidx, V, 0, Zd, d, _hip.SCORE_REFERENCE, sums2, None, scores)
    assert rel(scores, dots / D) < max(tol, 3e-7)
    # K1b (column-split runs): raw dots, denominators applied afterwards -- the same arithmetic, bit for bit
    late = torch.empty_like(scores)
    k.edge_score(rowptr, colidx, V, 0, Zd, d, _hip.SCORE_RAW_DOT, None, None, late)
    k.edge_score_finalize(rowptr, colidx, V, 0, _hip.SCORE_REFERENCE, sums2, None, late)
    assert torch.equal(late, scores)
    k.segment_softmax(rowptr, V, scores)
    P_ref = O.build_P_values(csr.rowptr, csr.colidx, Zf.double())
    assert rel(scores, P_ref) < max(tol, 3e-7)

    # per-edge (true cosine) mode
    k.edge_score(rowptr, colidx, V, 0, Zd, d, _hip.SCORE_PER_EDGE, None, sq, scores)
    k.edge_score(rowptr, colidx, V, 0, Zd, d, _hip.SCORE_RAW_DOT, None, None, late)
    k.edge_score_finalize(rowptr, colidx, V, 0, _hip.SCORE_PER_EDGE, None, sq, late)
    assert torch.equal(late, scores)
    k.segment_softmax(rowptr, V, scores)
    assert rel(scores, O.build_P_values(csr.rowptr, csr.colidx, Zf.double(), mode="per_edge")) < max(tol, 1e-6)


def test_segment_softmax_edge_cases(dev, k):
    deg = np.array([0, 1, 2, 63, 64, 65, 0, 0, 5000, 1, 129], dtype=np.int64)
    rowptr = np.zeros(len(deg) + 1, dtype=np.int64)
    np.cumsum(deg, out=rowptr[1:])
    for dtype in (torch.float32, torch.float64):
        vals = torch.randn(int(rowptr[-1]), dtype=dtype, generator=torch.Generator().manual_seed(1)) * 30
        vals[3] = 88.0                                         # large spread: max-subtraction matters
        ref = O.segment_softmax(rowptr, vals.double())
        out = vals.to(dev)
        k.segment_softmax(torch.from_numpy(rowptr).to(dev), len(deg), out)
        assert rel(out, ref) < (1e-6 if dtype == torch.float32 else 1e-14)
        sums = torch.zeros(len(deg), dtype=torch.float64).index_add_(
            0, torch.from_numpy(np.repeat(np.arange(len(deg)), deg)), out.cpu().double())
        assert torch.allclose(sums[deg > 0], torch.ones(int((deg > 0).sum()), dtype=torch.float64), atol=1e-5)


def run_sweep(k, rowptr, colidx, P, V, Zo, X, gamma, Zn, d, long_threshold, long_rows, partials, waves=16):
    k.spmm_update(rowptr, colidx, P, V, 0, Zo, X, gamma, Zn, d, long_threshold, partials)
    if long_rows is not None and long_rows.numel():
        k.spmm_update_long(rowptr, colidx, P, long_rows, waves, 0, Zo, X, gamma, Zn, d,
                           partials[k.spmm_partials_len(V, 0):])


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.bfloat16])
@pytest.mark.parametrize("d,pad", [(2, True), (16, True), (64, True), (128, True), (256, True), (300, True),
                                   (1433, True), (1433, False), (5, False)])
@pytest.mark.parametrize("long_threshold,waves", [(0, 16), (48, 16), (48, 4)])
def test_spmm_update_vs_oracle(dev, k, dtype, d, pad, long_threshold, waves):
    csr = ragged_csr(600, seed=d + 1, hubs=(64, 65, 200, 600, 1))
    V = csr.num_vertices
    acc = _hip.acc_dtype(dtype)
    gamma = 0.76
    X = synth.gaussian_X(V, d, seed=4).to(dtype)
    Zold = (synth.gaussian_X(V, d, seed=5) * 0.5).to(dtype)
    P = O.build_P_values(csr.rowptr, csr.colidx, synth.gaussian_X(V, 8, seed=6).double()).to(acc)
    mk = (lambda t: padded(t, dtype, dev)) if pad else (lambda t: t.to(dev).contiguous())
    Xd, Zo = mk(X), mk(Zold)
    Zn = torch.full_like(Zo, float("nan"))
    if pad:
        Zn[:, d:] = 0
    rowptr, colidx = torch.from_numpy(csr.rowptr).to(dev), torch.from_numpy(csr.colidx).to(dev)
    deg = np.diff(csr.rowptr)
    long_rows = torch.from_numpy(np.nonzero(deg > long_threshold)[0].astype(np.int32)).to(dev) \
        if long_threshold else None
    n_long = 0 if long_rows is None else long_rows.numel()
    partials = torch.full((k.spmm_partials_len(V, n_long),), float("nan"), dtype=torch.float64, device=dev)
    out = torch.zeros(1, dtype=torch.float64, device=dev)

    run_sweep(k, rowptr, colidx, P.to(dev), V, Zo, Xd, gamma, Zn, d, long_threshold, long_rows, partials, waves)
    k.reduce_partials(partials, partials.numel(), torch.zeros(k.reduce_ws_len(), dtype=torch.float64, device=dev), out)
    Z_ref, _ = O.sweep(csr.rowptr, csr.colidx, P.double(), X.double(), Zold.double(), gamma)
    got = Zn[:, :d].cpu()
    assert not torch.isnan(got.float()).any()
    assert rel(got, Z_ref) < TOL[dtype]
    sink = torch.from_numpy(deg == 0)
    assert torch.equal(got[sink], Zold[sink])                              # embedder.py:88-89: untouched rows
    if pad:
        assert float(Zn[:, d:].abs().sum()) == 0.0                         # pad columns stay zero
    delta_ref = (got.double() - Zold.double()).abs().sum()
    assert float(out) == pytest.approx(float(delta_ref), rel=1e-6 if dtype != torch.float64 else 1e-12)

    # bitwise reproducible: fixed-order partials, no atomics
    Zn2 = torch.zeros_like(Zn)
    partials2 = torch.zeros_like(partials)
    run_sweep(k, rowptr, colidx, P.to(dev), V, Zo, Xd, gamma, Zn2, d, long_threshold, long_rows, partials2, waves)
    assert torch.equal(Zn2[:, :d], Zn[:, :d]) and torch.equal(partials, partials2)


@pytest.mark.parametrize("case", range(60))
def test_k3_and_k1_random_shapes(dev, k, case):
    """60 seeded random cases (one test each: a failure names its case) -- V from 1 to 700 (fewer rows than a workgroup's block, than a wave's sub-waves;
    block boundaries), d from 1 to 320, all dtypes, degree mixes from all-empty to all-long, any long threshold --
    K3 with sinks untouched and K1 + fused softmax against the oracle.  Shapes the sub-wave kernels schedule
    differently (rows claimed per sub-wave) must not change a result."""
    rng = np.random.default_rng(2024 + case)
    if True:
        dtype = [torch.float32, torch.float64, torch.bfloat16][case % 3]
        V = int(rng.choice([1, 2, 3, 7, 8, 9, 31, 32, 33, 64, 100, 257, 700]))
        d = int(rng.choice([1, 3, 4, 8, 12, 16, 24, 32, 40, 64, 100, 128, 160, 256, 320]))
        style = case % 4
        max_deg = [0, 3, 40, min(V, 120)][style]
        hubs = () if style < 2 else tuple(int(h) for h in rng.integers(1, V + 1, size=3))
        csr = ragged_csr(V, seed=1000 + case, max_deg=min(max_deg, V), hubs=hubs, empty_frac=[1.0, 0.3, 0.2, 0.0][style])
        T = int(rng.choice([0, 1, 8, 48, 1000]))
        acc, gamma = _hip.acc_dtype(dtype), 0.6
        X = synth.gaussian_X(V, d, seed=case).to(dtype)
        Zold = (synth.gaussian_X(V, d, seed=case + 7) * 0.5).to(dtype)
        Xd, Zo = padded(X, dtype, dev), padded(Zold, dtype, dev)
        rowptr, colidx = torch.from_numpy(csr.rowptr).to(dev), torch.from_numpy(csr.colidx).to(dev)
        if colidx.numel() == 0:                # the ABI wants real pointers even when rowptr says "no edges"
            colidx = torch.zeros(1, dtype=torch.int32, device=dev)
        deg = np.diff(csr.rowptr)
        long_rows = torch.from_numpy(np.nonzero(deg > T)[0].astype(np.int32)).to(dev) if T else None
        n_long = 0 if long_rows is None else long_rows.numel()
        tag = f"case {case}: V={V} d={d} {dtype} style={style} T={T} E={csr.num_edges}"
        # K1 + softmax (reference mode needs nonzero norms: Zold is random)
        sq = torch.empty(V, dtype=acc, device=dev)
        k.row_sqnorm(Zo, d, sq)
        sums2 = torch.zeros(2, dtype=torch.float64, device=dev)
        k.degree_weighted_sums(sq, rowptr, torch.from_numpy(csr.indeg()).to(dev), V,
                               torch.zeros(k.reduce_ws_len(), dtype=torch.float64, device=dev), sums2)
        P = torch.full((max(csr.num_edges, 1),), float("nan"), dtype=acc, device=dev)
        if csr.num_edges:
            k.edge_score(rowptr, colidx, V, 0, Zo, d, _hip.SCORE_REFERENCE, sums2, None, P, T,
                         long_rows if n_long else None, fuse_softmax=True)
            P_ref = O.build_P_values(csr.rowptr, csr.colidx, Zold.to(acc).double())
            assert rel(P[:csr.num_edges], P_ref) < (max(TOL[dtype], 1e-6) if dtype != torch.bfloat16 else 1e-4), tag
        else:
            P.zero_()
        # K3
        Zn = Zo.clone()
        partials = torch.zeros(k.spmm_partials_len(V, n_long), dtype=torch.float64, device=dev)
        k.spmm_update(rowptr, colidx, P, V, 0, Zo, Xd, gamma, Zn, d, T, partials, sinks_untouched=True)
        if n_long:
            k.spmm_update_long(rowptr, colidx, P, long_rows, 16, 0, Zo, Xd, gamma, Zn, d,
                               partials[k.spmm_partials_len(V, 0):])
        Z_ref, _ = O.sweep(csr.rowptr, csr.colidx, P[:csr.num_edges].cpu().double(), X.double(), Zold.double(), gamma)
        got = Zn[:, :d].cpu()
        assert rel(got, Z_ref) < TOL[dtype], tag
        assert float(Zn[:, d:].abs().sum()) == 0.0, tag
        out = torch.zeros(1, dtype=torch.float64, device=dev)
        k.reduce_partials(partials, partials.numel(), torch.zeros(k.reduce_ws_len(), dtype=torch.float64, device=dev), out)
        want = float((got.double() - Zold.double()).abs().sum())
        assert float(out) == pytest.approx(want, rel=1e-6, abs=1e-12), tag


@pytest.mark.parametrize("case", range(48))
def test_class_pass_random_engines(dev, case):
    """48 seeded random graphs (one test each: a failure names its case) through the WHOLE engine with tiny class thresholds, so that most rows take the
    XCD-affine pass (K1 and K3): V from 1 to 900, d from 1 to 300, fp32 / fp64 / bf16, 1 or 3 launch blocks, natural or
    hot-first order, rows whose edges all fall into one class, dense rows, empty rows -- build_P (reference and
    per-edge) and three sweeps against the oracle."""
    rng = np.random.default_rng(7700 + case)
    if True:
        dtype = [torch.float32, torch.float64, torch.bfloat16][case % 3]
        V = int(rng.choice([1, 2, 7, 8, 9, 33, 64, 100, 257, 900]))
        d = int(rng.choice([1, 2, 4, 8, 16, 24, 64, 100, 128, 256, 300]))
        deg = rng.integers(0, min(V, 12) + 1, size=V)
        deg[rng.random(V) < 0.2] = 0
        for h in rng.integers(1, V + 1, size=3):
            deg[int(rng.integers(V))] = int(h)
        cols = []
        for r, kdeg in enumerate(deg):
            if case % 8 == 5 and kdeg and V >= 64:          # every neighbour in ONE 8-row block of the table order
                pool = np.arange(0, V, 1)
                c = np.sort(rng.choice(pool[pool % 64 < 8], size=min(kdeg, int((pool % 64 < 8).sum())), replace=False))
                deg[r] = c.size
            else:
                c = np.sort(rng.choice(V, size=kdeg, replace=False))
            cols.append(c)
        rowptr = np.zeros(V + 1, dtype=np.int64)
        np.cumsum(deg, out=rowptr[1:])
        csr = HostCSR(V, rowptr, np.concatenate(cols + [np.empty(0, int)]).astype(np.int32))
        acc = _hip.acc_dtype(dtype)
        X = synth.gaussian_X(V, d, seed=case).to(dtype)
        ct, chunk = int(rng.choice([1, 2, 5, 8])), int(rng.choice([64, 128, 256]))
        mode = "per_edge" if case % 2 else "reference"
        phases, pt = int(rng.choice([1, 2, 4, 8])), int(rng.choice([2, 6, 20]))     # heavy rows phased in time as well
        eng = SweepEngine(csr, X, dev, chunks=1 if case % 4 else 3, hot_rows_first=bool(case % 3), cosine_mode=mode,
                          class_threshold=ct, class_chunk=chunk, class_phases=phases, phase_threshold=pt)
        tag = (f"case {case}: V={V} d={d} {dtype} class_threshold={ct} chunk={chunk} phases={phases}>{pt} "
               f"E={csr.num_edges} {mode}")
        assert sum(0 if c is None else c[0].numel() for c in eng.class_rows) == int((deg > ct).sum()), tag
        Xf = X.to(acc).double()
        if csr.num_edges == 0:
            return
        eng.build_P()
        P_or = O.build_P_values(csr.rowptr, csr.colidx, Xf, mode=mode)
        tol_p = 1e-12 if dtype == torch.float64 else 2e-5 if dtype == torch.bfloat16 else 5e-6
        assert rel(eng.P_global(), P_or) < tol_p, tag
        Z, P_used = Xf.clone(), eng.P_global().double()
        for _ in range(3):
            delta = eng.sweep(0.6)
            Z_next, d_or = O.sweep(csr.rowptr, csr.colidx, P_used, Xf, Z, 0.6)
            got = eng.get_Z().double()
            assert rel(got, Z_next) < TOL[dtype], tag
            assert delta == pytest.approx(float((got - Z).abs().sum()), rel=1e-5 if dtype != torch.bfloat16 else 2e-2,
                                          abs=1e-9), tag
            Z = got                                  # follow the GPU's (bf16-rounded) trajectory
        del eng


@pytest.mark.parametrize("dtype,d,tiles", [(torch.float32, 256, 2), (torch.float32, 100, 3), (torch.float64, 64, 2),
                                           (torch.bfloat16, 512, 2), (torch.float32, 1433, 4)])
def test_column_tiles_on_the_gpu(dev, dtype, d, tiles):
    """SweepEngine(column_tiles=T) with the real kernels: the same tables swept as T column ranges (strided views: rows of
    T-th the width, embedder.py:92 column by column) == the oracle and the untiled engine up to summation order; bitwise
    repeatable; build_P, snapshot / outer delta and get_Z see whole rows."""
    csr = ragged_csr(6000, seed=21, max_deg=40, hubs=(5000, 900, 300, 129, 65, 64, 33))
    X = synth.gaussian_X(6000, d, seed=5).to(dtype)
    acc = _hip.acc_dtype(dtype)
    kw = dict(class_threshold=32, class_chunk=64)
    plain = SweepEngine(csr, X, dev, column_tiles=1, **kw)
    tiled = SweepEngine(csr, X, dev, column_tiles=tiles, **kw)
    again = SweepEngine(csr, X, dev, column_tiles=tiles, **kw)
    assert len(tiled.tiles) == tiles and tiled.tiles[0][0] == 0 and tiled.tiles[-1][1] == d
    assert all(a[1] == b[0] for a, b in zip(tiled.tiles, tiled.tiles[1:]))
    for eng in (plain, tiled, again):
        eng.build_P()
    P_or = O.build_P_values(csr.rowptr, csr.colidx, X.to(acc).double())
    assert rel(tiled.P_global(), P_or) < (1e-12 if dtype == torch.float64 else 2e-5 if dtype == torch.bfloat16 else 5e-6)
    Z = X.to(acc).double()
    P_used = tiled.P_global().double()
    tiled.snapshot()
    for sweep in range(3):
        dt, dp, da = tiled.sweep(0.7), plain.sweep(0.7), again.sweep(0.7)
        assert dt == da and torch.equal(tiled.Zcur, again.Zcur)                 # bitwise repeatable
        Zn, d_or = O.sweep(csr.rowptr, csr.colidx, P_used, X.to(acc).double(), Z, 0.7)
        got = tiled.get_Z().double()
        assert rel(got, Zn) < TOL[dtype] and rel(got, plain.get_Z().double()) < TOL[dtype]
        assert dt == pytest.approx(float((got - Z).abs().sum()), rel=1e-5 if dtype != torch.bfloat16 else 2e-2)
        assert dt == pytest.approx(dp, rel=1e-5 if dtype != torch.bfloat16 else 2e-2)
        Z = got
    moved = tiled.distance_from_snapshot()
    assert moved == pytest.approx(float((Z - X.to(acc).double()).abs().sum()), rel=1e-5 if dtype != torch.bfloat16 else 2e-2)
    assert_norms_are_k0s(tiled, "tiled engine, after the outer-delta pass")


def test_table_beyond_cache_hint_changes_no_bit(dev, k):
    """CLANE_SPMM_TABLE_BEYOND_CACHE picks other instances of the 512-byte-row kernels (fewer row loads in flight, more
    waves) -- a speed hint: Z, deltas and partials are the same bits with and without it, for the instances it selects
    (fp32, d = 128: two rows per instruction) and for those that ignore it."""
    csr = ragged_csr(8000, seed=31, max_deg=50, hubs=(7000, 800, 129, 65))
    for dtype, d in ((torch.float32, 128), (torch.float32, 100), (torch.float32, 256), (torch.bfloat16, 128)):
        X = synth.gaussian_X(8000, d, seed=2).to(dtype)
        a = SweepEngine(csr, X, dev, class_threshold=32, class_chunk=64)
        b = SweepEngine(csr, X, dev, class_threshold=32, class_chunk=64)
        assert not a.beyond_cache and not a.kernel_config()["fewer_loads_in_flight"]
        b.beyond_cache = True                                  # (a small table: forced, before the launch lists are built)
        assert b.kernel_config()["fewer_loads_in_flight"] == ((dtype == torch.float32 and d in (128, 100)) or dtype == torch.bfloat16)
        for eng in (a, b):
            eng.build_P()
        for _ in range(3):
            assert a.sweep(0.7) == b.sweep(0.7)
            assert torch.equal(a.Zcur, b.Zcur) and torch.equal(a.partials, b.partials)


def test_set_cosine_mode_switches_the_scores_of_a_live_engine(dev):
    """SweepEngine.set_cosine_mode (bench.py's non-degenerate full-size check of K1 uses it): the same engine scores in
    per-edge mode, then in reference mode again -- each time the P of a fresh engine of that mode, bit for bit, and the
    oracle's (similarity.py:35-37 vs the cosine its docstring describes); sweeps refuse to run on a P of the other mode."""
    csr = synth.rmat_csr(30_000, 600_000, seed=3)
    X = synth.gaussian_X(30_000, 64, seed=4)
    eng = SweepEngine(csr, X, dev)
    eng.build_P()
    P_ref = eng.P.clone()
    eng.set_cosine_mode("per_edge")
    with pytest.raises(RuntimeError, match="before build_P"):
        eng.sweep(0.5)
    eng.build_P()
    fresh = SweepEngine(csr, X, dev, cosine_mode="per_edge")
    fresh.build_P()
    assert torch.equal(eng.P, fresh.P) and not torch.equal(eng.P, P_ref)
    assert rel(eng.P_global(), O.build_P_values(csr.rowptr, csr.colidx, X, mode="per_edge")) < 5e-6
    eng.set_cosine_mode("reference")
    eng.build_P()
    assert torch.equal(eng.P, P_ref)
    with pytest.raises(ValueError, match="cosine_mode"):
        eng.set_cosine_mode("cosine")


def assert_norms_are_k0s(eng, tag=""):
    """The squared row norms the outer-delta pass leaves behind (SweepEngine.sq_pp) are BIT FOR BIT what
    row_sqnorm_kernel (K0) computes from the same table -- for every owned row, sinks included."""
    want = torch.zeros_like(eng.sq_pp[eng.cur])
    for b in eng.blocks:
        eng.k.row_sqnorm(eng._zrows(eng.Zcur, b), eng.d, want[eng._rows(b)])
    got = eng.sq_pp[eng.cur]
    assert torch.equal(got, want), (tag, int((got != want).sum()), float((got - want).abs().max()))


@pytest.mark.parametrize("dtype,d", [(torch.float32, 256), (torch.float32, 100), (torch.float32, 1433), (torch.float64, 64),
                                     (torch.bfloat16, 128), (torch.bfloat16, 24), (torch.float32, 2)])
@pytest.mark.parametrize("kw", [dict(), dict(class_threshold=0), dict(class_threshold=0, split_hubs=False, chunks=3),
                                dict(class_threshold=16, class_chunk=64, chunks=2), dict(long_threshold=8, hub_threshold=40,
                                                                                         class_threshold=0)])
def test_row_norms_are_bitwise_k0(dev, dtype, d, kw):
    """build_P's row norms (similarity.py:37) come from K0 only when nobody has left them behind: the outer-delta pass
    (l1_between: it reads every row of the new Z anyway) leaves norms that are the SAME BITS as K0's on the table they
    belong to, whichever K3 kernels wrote it.  The sweeps leave nothing behind; the third table appears with the first
    snapshot()."""
    csr = ragged_csr(5000, seed=11, max_deg=30, hubs=(4500, 700, 129, 65, 64, 5000, 300, 33))
    X = synth.gaussian_X(5000, d, seed=3).to(dtype)
    eng = SweepEngine(csr, X, dev, **kw)
    eng.build_P()
    assert eng.sq_ok[eng.cur] and len(eng.Zbuf) == 2
    assert_norms_are_k0s(eng, "after load")
    eng.sweep(0.7)
    eng.snapshot()                          # the pinned table is never a destination: three tables rotate from here on
    assert len(eng.Zbuf) == 3 and len(eng.sq_pp) == 3 and torch.equal(eng.Zbuf[2], eng.Zcur)
    for _ in range(2):
        eng.sweep(0.7)
    assert not eng.sq_ok[eng.cur]           # a sweep's destination starts without norms ...
    moved = eng.distance_from_snapshot()
    assert moved > 0 and eng.sq_ok[eng.cur]  # ... the outer-delta pass leaves them
    assert_norms_are_k0s(eng, "after the outer-delta pass")
    eng.build_P()
    P_l1 = eng.P.clone()
    eng.sq_ok[eng.cur] = False
    eng.build_P()                           # K0 again: the same P, bit for bit
    assert torch.equal(P_l1, eng.P)


def test_spmm_row_block_with_row0_offset(dev, k):
    """A rank's row block: local rowptr/X/Z_new, global columns, row0 != 0."""
    csr = ragged_csr(400, seed=9, hubs=(300,))
    V, d, gamma, r0, n = 400, 128, 0.5, 150, 200
    X, Zold = synth.gaussian_X(V, d, seed=1), synth.gaussian_X(V, d, seed=2)
    P = O.build_P_values(csr.rowptr, csr.colidx, X.double()).float()
    Z_ref, _ = O.sweep(csr.rowptr, csr.colidx, P.double(), X.double(), Zold.double(), gamma)
    rp = torch.from_numpy(csr.rowptr).to(dev)
    Zn = torch.zeros(n, d, device=dev)
    partials = torch.zeros(k.spmm_partials_len(n, 0), dtype=torch.float64, device=dev)
    k.spmm_update(rp[r0:], torch.from_numpy(csr.colidx).to(dev), P.to(dev), n, r0, Zold.to(dev),
                  X[r0:r0 + n].to(dev), gamma, Zn, d, 0, partials)
    assert rel(Zn, Z_ref[r0:r0 + n]) < 2e-6
    with pytest.raises(_hip.ClaneHipError, match="alias"):
        Zo = Zold.to(dev)
        k.spmm_update(rp, torch.from_numpy(csr.colidx).to(dev), P.to(dev), V, 0, Zo, X.to(dev), gamma, Zo, d, 0,
                      torch.zeros(k.spmm_partials_len(V, 0), dtype=torch.float64, device=dev))


def test_l1_distance(dev, k):
    """sum|A - B| (embedder.py:60) and, from the same pass, the squared norm of every row of A -- bit for bit what
    row_sqnorm (K0) computes: padded and unpadded leading dimensions, one and several column tiles, a row count that
    leaves the last wave half empty."""
    for dtype, d, pad in [(torch.float32, 256, True), (torch.float64, 7, False), (torch.bfloat16, 128, True),
                          (torch.float32, 1433, True), (torch.float32, 100, True), (torch.bfloat16, 24, True),
                          (torch.float32, 2, True), (torch.float64, 64, True), (torch.float32, 37, False)]:
        A, B = synth.gaussian_X(321, d, seed=1).to(dtype), synth.gaussian_X(321, d, seed=2).to(dtype)
        Ad, Bd = padded(A, dtype, dev, None if pad else d), padded(B, dtype, dev, None if pad else d)
        ws = torch.zeros(k.reduce_ws_len(), dtype=torch.float64, device=dev)
        out = torch.zeros(1, dtype=torch.float64, device=dev)
        k.l1_distance(Ad, Bd, d, ws, out)
        want = float((A.double() - B.double()).abs().sum())
        assert float(out) == pytest.approx(want, rel=1e-6)
        acc = _hip.acc_dtype(dtype)
        sq, sq_k0 = torch.full((321,), -1.0, dtype=acc, device=dev), torch.zeros(321, dtype=acc, device=dev)
        out2 = torch.zeros(1, dtype=torch.float64, device=dev)
        k.l1_distance(Ad, Bd, d, ws, out2, sq_a=sq)
        k.row_sqnorm(Ad, d, sq_k0)
        assert torch.equal(out, out2) and torch.equal(sq, sq_k0), (dtype, d, pad)
        assert rel(sq, A.to(acc).double().pow(2).sum(1)) < TOL[dtype]


# ---- end to end through the reference surface, against the goldens ------------------------------------
@pytest.mark.parametrize("name", KARATE_LIKE)
def test_graph_build_P_matches_reference(tmp_path, name):
    gold, g = graph_from_golden(tmp_path, name)
    P = g.build_P(CosineSimilarity())
    assert P.is_coalesced() and P.dtype == g.X.dtype
    np.testing.assert_array_equal(P.indices().numpy(), gold["A_indices"])
    np.testing.assert_allclose(P.values().numpy(), gold["P0_values"], rtol=1e-5, atol=1e-7)
    dense = P.to_dense()
    assert dense.shape == (len(g), len(g))
    assert dense.sum(1).max().item() == pytest.approx(1, abs=1e-3)          # reference test_graph.py:47-48
    if (np.diff(g.csr.rowptr) == 0).any():
        assert dense.sum(1).min().item() == pytest.approx(0, abs=1e-3)


@pytest.mark.parametrize("name", KARATE_LIKE)
@pytest.mark.parametrize("chunks", [1, 3])
def test_embedder_iterate_matches_reference(tmp_path, name, chunks):
    gold, g = graph_from_golden(tmp_path, name)
    eng = g.engine(chunks=chunks, shuffle=chunks > 1)
    if name.startswith(("g11", "g12")):     # fixtures with hub rows, made by the real reference: taken by the class pass
        assert eng.class_threshold == (256 if name.startswith("g11") else 64)
        assert sum(0 if c is None else c[0].numel() for c in eng.class_rows) == (1 if name.startswith("g11") else 2)
    emb = Embedder(g, CosineSimilarity(), torch.device("cpu"), gamma=float(gold["gamma"]),
                   tolerence=int(gold["tolerence"]), save_history=True, verbose=False)
    emb.iterate()
    Z = g.Z
    assert Z.dtype == g.X.dtype and not Z.is_cuda
    tol = 1e-12 if g.X.dtype == torch.float64 else 1e-5
    assert O.rel_l2(Z, torch.from_numpy(gold["Z_final"])) < tol
    assert O.rel_l2(emb.history["Z"][0][0], torch.from_numpy(gold["Z_sweep1"])) < tol
    assert O.rel_l2(emb.history["Z"][0][-1], torch.from_numpy(gold["Z_prop1"])) < tol
    assert (g.Z - g.X).abs().sum() != 0                                     # reference test_embedder.py:29
    sink = np.diff(g.csr.rowptr) == 0
    np.testing.assert_array_equal(Z.numpy()[sink], gold["X"][sink])


def test_corashape_literal_sweep_golden(dev):
    g = load_golden("g8_corashape.npz")
    V, d = int(g["V"]), int(g["d"])
    X = torch.zeros(V, d)
    X[torch.from_numpy(g["X_nz_row"].astype(np.int64)), torch.from_numpy(g["X_nz_col"].astype(np.int64))] = 1.0
    rowptr, colidx = O.build_csr(V, g["src"], g["dst"])
    eng = SweepEngine(HostCSR(V, rowptr, colidx), X, dev)
    eng.build_P()
    np.testing.assert_allclose(eng.P_global().numpy(), g["P_values"], rtol=1e-5)
    delta = eng.sweep(0.76)
    Z1 = eng.get_Z()
    np.testing.assert_allclose(Z1[:24].numpy(), g["Z1_head"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(Z1.norm(dim=1).numpy(), g["Z1_rownorm"], rtol=1e-5)
    assert delta == pytest.approx(float(g["delta1"]), rel=1e-5)


def test_corashape_converged_embeddings_config1(tmp_path):
    """BASELINE config 1: Cora-shaped graph (2708 / 5429 / d=1433 BoW) to convergence, within 1e-4."""
    g = load_golden("g8_corashape.npz")
    V, d = int(g["V"]), int(g["d"])
    X = torch.zeros(V, d)
    X[torch.from_numpy(g["X_nz_row"].astype(np.int64)), torch.from_numpy(g["X_nz_col"].astype(np.int64))] = 1.0
    root = write_data_root(tmp_path / "cora", range(V), g["src"], g["dst"], X.numpy())
    graph = Graph(root)
    assert graph.csr.num_edges == 5429
    emb = Embedder(graph, CosineSimilarity(), torch.device("cuda"), gamma=0.76, tolerence=10, verbose=False)
    emb.iterate()
    orc = O.OracleEmbedder(graph.csr.rowptr, graph.csr.colidx, X, gamma=0.76, tolerence=10, plain_c=True)
    assert O.rel_l2(graph.Z, orc.iterate()) < 1e-5


def test_cli_end_to_end(tmp_path, karate_root, monkeypatch):
    from clane_amd.__main__ import embedding, get_parser
    cfg = tmp_path / "config.yaml"
    cfg.write_text("graph:\n  embedding_dim: 2\n\nsimilarity:\n  method: \"CosineSimilarity\"\n  kwargs:\n"
                   "    foo: \"bar\"\n\nembedder:\n  gamma: 0.76\n  tolerence: 10\n")
    for n_run, argv in enumerate((["--data_root", str(karate_root)], ["embedding", "--data_root", str(karate_root)])):
        out = tmp_path / f"test_output{n_run}"       # no C.npy: each run draws its own X and writes its own history
        args = get_parser().parse_args(argv + ["--output_root", str(out), "--config_file", str(cfg),
                                               "--save_history"])
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            embedding(args)
        gold = load_golden("g9_cli.npz")
        Z = np.load(out / "Z.npy")
        assert Z.shape == (34, 2) and np.load(out / "0" / "Z_0.npy").shape == (34, 2)   # reference test_cli.py:31-36
        listing = sorted(str(p.relative_to(out)) for p in out.rglob("*.npy"))
        assert "Z.npy" in listing and "0/Z_0.npy" in listing
        # streamed while the sweeps ran: every sweep of every outer round is there, the last one is the result
        rounds = sorted(int(p.name) for p in out.iterdir() if p.is_dir())
        assert rounds == list(range(len(rounds))) and len(rounds) >= 3
        last = max((out / str(rounds[-1])).glob("Z_*.npy"), key=lambda p: int(p.stem.split("_")[1]))
        np.testing.assert_array_equal(np.load(last), Z)
        text = buf.getvalue()
        for line in ("[Embedding]", "Graph Loaded.", " - 34 vertices", " - 78 edges", "Saving the results."):
            assert line in text
        assert set(gold["parser_dests"]) <= {a.dest for a in get_parser()._actions}
    # --init_Z (extension): resuming from the converged embeddings (content stored in C.npy this time, so both
    # runs see the same X) reproduces them and stops after the minimum number of rounds
    k4 = load_golden("g4_karate_d2.npz")
    kc = load_golden("g2_karate_csr.npz")
    root_c = write_data_root(tmp_path / "with_content", kc["vertex_ids"], kc["edge_src"], kc["edge_dst"], k4["X"])
    out1, out2 = tmp_path / "first", tmp_path / "resumed"
    with contextlib.redirect_stdout(io.StringIO()):
        embedding(get_parser().parse_args(["--data_root", str(root_c), "--output_root", str(out1), "--config_file",
                                           str(cfg), "--save_history"]))
        embedding(get_parser().parse_args(["--data_root", str(root_c), "--output_root", str(out2), "--config_file",
                                           str(cfg), "--init_Z", str(out1 / "Z.npy"), "--save_history"]))
    Z1 = np.load(out1 / "Z.npy")
    assert np.linalg.norm(Z1 - k4["Z_final"]) <= 1e-5 * np.linalg.norm(Z1)
    assert np.linalg.norm(np.load(out2 / "Z.npy") - Z1) <= 1e-5 * np.linalg.norm(Z1)
    assert len([p for p in out2.iterdir() if p.is_dir()]) <= len([p for p in out1.iterdir() if p.is_dir()])
    with pytest.raises(ValueError, match="--init_Z"):
        np.save(tmp_path / "wrong.npy", np.zeros((3, 2), dtype=np.float32))
        embedding(get_parser().parse_args(["--data_root", str(karate_root), "--output_root", str(out2), "--config_file",
                                           str(cfg), "--init_Z", str(tmp_path / "wrong.npy")]))
    with pytest.raises(FileNotFoundError):
        embedding(get_parser().parse_args(["--data_root", str(karate_root), "--output_root", str(out),
                                           "--config_file", str(tmp_path / "missing.yaml")]))
    bad = tmp_path / "bad.yaml"
    bad.write_text("graph:\n  embedding_dim: 2\nsimilarity:\n  method: \"Nope\"\n  kwargs: {}\nembedder: {}\n")
    with pytest.raises(AttributeError, match="not found"):
        embedding(get_parser().parse_args(["--data_root", str(karate_root), "--output_root", str(out),
                                           "--config_file", str(bad)]))


# ---- scale: config 2 shape (R-MAT 200k / 4M / d=128) directly against the oracle ------------------------
def test_rmat_200k_parity_and_properties(dev):
    V, E, d, gamma = 200_000, 4_000_000, 128, 0.76
    csr = synth.rmat_csr(V, E, seed=1)
    assert csr.num_edges == E and int(np.diff(csr.rowptr).max()) > 1024       # has hub rows -> long-row pass
    X = synth.gaussian_X(V, d, seed=2)
    eng = SweepEngine(csr, X, dev, long_threshold=64, hub_threshold=256, class_threshold=0)   # the three row kernels
    assert eng.hub_rows[0] is not None and eng.mid_rows[0] is not None
    eng.build_P()
    P_or = O.build_P_values(csr.rowptr, csr.colidx, X)
    P_gpu = eng.P_global()
    assert rel(P_gpu, P_or) < 1e-5
    # property: every non-empty row of P sums to 1
    rows = torch.from_numpy(np.repeat(np.arange(V), np.diff(csr.rowptr))).to(dev)
    rs = torch.zeros(V, dtype=torch.float64, device=dev).index_add_(0, rows, P_gpu.double().to(dev))
    nz = torch.from_numpy(np.diff(csr.rowptr) > 0).to(dev)
    assert float((rs[nz] - 1).abs().max()) < 1e-5 and float(rs[~nz].abs().max()) == 0
    Z = X.clone()
    Ps = O.as_sparse(csr.rowptr, csr.colidx, P_or)
    for _ in range(3):
        delta = eng.sweep(gamma)
        Z, d_or = O.sweep(csr.rowptr, csr.colidx, P_or, X, Z, gamma, Ps)
        assert delta == pytest.approx(float(d_or), rel=1e-4)
    assert O.rel_l2(eng.get_Z(), Z) < 1e-5
    # property (size-independent): the sweep is affine in Z -> S(a) - S(b) = gamma * P (a - b)
    a = eng.get_Z()
    eng.set_Z(a * 0.5)
    eng.P_valid = True
    eng.sweep(gamma)
    half = eng.get_Z()
    eng.set_Z(a)
    eng.P_valid = True
    eng.sweep(gamma)
    full = eng.get_Z()
    lin = (full - X) - 2 * (half - X)
    nzc = torch.from_numpy(np.diff(csr.rowptr) > 0)
    assert float(lin[nzc].abs().max()) < 1e-4 * float(full.abs().max())


def test_spmm_sinks_untouched_flag(dev, k):
    """CLANE_SPMM_SINKS_UNTOUCHED: rows without out-edges are neither read nor written."""
    csr = ragged_csr(300, seed=11, empty_frac=0.5)
    V, d, gamma = 300, 256, 0.76
    X, Zold = synth.gaussian_X(V, d, seed=1), synth.gaussian_X(V, d, seed=2)
    P = O.build_P_values(csr.rowptr, csr.colidx, X.double()).float().to(dev)
    rowptr, colidx = torch.from_numpy(csr.rowptr).to(dev), torch.from_numpy(csr.colidx).to(dev)
    sink = torch.from_numpy(np.diff(csr.rowptr) == 0)
    out = {}
    for flag in (False, True):
        Zn = torch.full((V, d), 7.0, device=dev)
        partials = torch.zeros(k.spmm_partials_len(V, 0), dtype=torch.float64, device=dev)
        k.spmm_update(rowptr, colidx, P, V, 0, Zold.to(dev), X.to(dev), gamma, Zn, d, 0, partials, sinks_untouched=flag)
        out[flag] = (Zn.cpu(), partials.cpu())
    assert torch.equal(out[False][0][sink], Zold[sink])                 # default: copied (embedder.py:88-89)
    assert torch.equal(out[True][0][sink], torch.full((int(sink.sum()), d), 7.0))   # flag: left alone
    assert torch.equal(out[False][0][~sink], out[True][0][~sink])
    assert torch.equal(out[False][1], out[True][1])                     # sinks contribute 0 to the delta either way


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.bfloat16])
@pytest.mark.parametrize("d,pad", [(2, True), (16, True), (128, True), (256, True), (1433, True), (37, False)])
def test_edge_score_fused_softmax(dev, k, dtype, d, pad):
    """K1 with CLANE_SCORE_FUSE_SOFTMAX (every row, listed long rows included) == K1 raw + K2 over every row."""
    csr = ragged_csr(400, seed=d + 7, max_deg=70, hubs=(64, 65, 200, 1))
    V, acc = csr.num_vertices, _hip.acc_dtype(dtype)
    Zc = synth.gaussian_X(V, d, seed=5).to(dtype)
    Zd = padded(Zc, dtype, dev) if pad else Zc.to(dev).contiguous()
    rowptr, colidx = torch.from_numpy(csr.rowptr).to(dev), torch.from_numpy(csr.colidx).to(dev)
    deg = np.diff(csr.rowptr)
    sq = torch.empty(V, dtype=acc, device=dev)
    k.row_sqnorm(Zd, d, sq)
    for lt, lr in ((0, None), (32, torch.from_numpy(np.nonzero(deg > 32)[0].astype(np.int32)).to(dev))):
        two_pass = torch.zeros(csr.num_edges, dtype=acc, device=dev)
        k.edge_score(rowptr, colidx, V, 0, Zd, d, _hip.SCORE_PER_EDGE, None, sq, two_pass, lt, lr)
        k.segment_softmax(rowptr, V, two_pass)
        fused = torch.zeros_like(two_pass)
        k.edge_score(rowptr, colidx, V, 0, Zd, d, _hip.SCORE_PER_EDGE, None, sq, fused, lt, lr, fuse_softmax=True)
        # same scores, same softmax up to the reduction order (in-register / running max-sum per wave combined by
        # the workgroup / K2's three passes)
        assert rel(fused, two_pass) < (1e-14 if dtype == torch.float64 else 3e-7)
        assert float((fused - two_pass).abs().max()) < (1e-14 if dtype == torch.float64 else 2e-7)
    ref = O.build_P_values(csr.rowptr, csr.colidx, Zc.to(acc).double(), mode="per_edge")
    assert rel(fused, ref) < (1e-13 if dtype == torch.float64 else 5e-6)


def test_bf16_storage_end_to_end(tmp_path):
    """Graph(dtype="bfloat16"): bf16 Z/X in HBM, fp32 accumulate and P; converges next to the fp32 reference."""
    gold, _ = graph_from_golden(tmp_path, "g5_symkarate_d16_g0.76.npz")
    g = Graph(tmp_path / "g5_symkarate_d16_g0.76.npz", dtype="bfloat16")
    assert g.X.dtype == torch.bfloat16
    P = g.build_P(CosineSimilarity())
    assert P.dtype == torch.float32
    np.testing.assert_allclose(P.values().numpy(), gold["P0_values"], rtol=2e-2)
    emb = Embedder(g, CosineSimilarity(), torch.device("cuda"), gamma=float(gold["gamma"]), tolerence=10,
                   verbose=False, max_sweeps=300)
    emb.iterate()
    Z = g.Z
    assert Z.dtype == torch.bfloat16 and len(emb.outer_deltas) < 40
    assert O.rel_l2(Z.float(), torch.from_numpy(gold["Z_final"])) < 2e-2


def test_powerlaw_generator_and_bf16_sweep(dev):
    csr = synth.powerlaw_csr(100_000, 2_000_000, seed=5)
    deg = np.diff(csr.rowptr)
    assert csr.num_edges == 2_000_000 and deg.max() > 1000 and (np.diff(csr.colidx.astype(np.int64))[
        np.diff(np.repeat(np.arange(100_000), deg)) == 0] > 0).all()          # sorted, unique within rows
    X = synth.gaussian_X(100_000, 128, seed=6).to(torch.bfloat16)
    eng = SweepEngine(csr, X, dev)
    eng.build_P()
    P_or = O.build_P_values(csr.rowptr, csr.colidx, X.float())
    assert rel(eng.P_global(), P_or) < 1e-5
    delta = eng.sweep(0.76)
    Z1, d_or = O.sweep(csr.rowptr, csr.colidx, P_or, X.float(), X.float(), 0.76)
    assert O.rel_l2(eng.get_Z().float(), Z1) < 8e-3 and delta == pytest.approx(float(d_or), rel=2e-2)


@pytest.mark.parametrize("exchange,world,fused,d", [
    ("columns", 4, True, 256), ("columns", 3, True, 256), ("columns", 4, True, 6), ("columns", 2, True, 100),
    ("columns", 2, True, 132),        # 33 packs: slices of 32 and 16 lanes per row, ONE class-sorted edge order (r03 fix)
    ("halo", 4, True, 256), ("halo", 3, True, 256), ("halo", 4, False, 256), ("allgather", 4, True, 256),
    ("halo_p2p", 4, True, 256), ("halo_p2p", 3, True, 100),
    # the divisions of the driver's N = 8 record at that world size, with the real kernels
    ("allgather_all", 8, True, 64), ("allgather", 8, True, 64), ("halo", 8, True, 64), ("columns", 8, True, 64)])
def test_partitioned_engine_with_real_kernels_on_one_gpu(dev, exchange, world, fused, d):
    """W ranks as threads of this process, all on cuda:0, collectives through tests/thread_comm.py: the column
    split (even, uneven 64 packs / 3, ranks without columns at d=6, ragged d=100) and the halo / all-gather row
    layouts, chunking, relabelled CSR and send-buffer packing run with the real HIP kernels."""
    import threading
    from .thread_comm import ThreadWorld
    V, E, gamma = 20_000, 300_000, 0.76
    csr = synth.rmat_csr(V, E, seed=9)
    X = synth.gaussian_X(V, d, seed=10)
    P_or = O.build_P_values(csr.rowptr, csr.colidx, X, mode="per_edge" if d == 100 else "reference")
    Z_or, deltas_or = X.clone(), []
    Ps = O.as_sparse(csr.rowptr, csr.colidx, P_or)
    for _ in range(3):
        Z_or, dl = O.sweep(csr.rowptr, csr.colidx, P_or, X, Z_or, gamma, Ps)
        deltas_or.append(float(dl))
    shared, results, errors = ThreadWorld(world), [None] * world, []

    def run(rank):
        try:
            with torch.cuda.device(dev):
                eng = SweepEngine(csr, X, dev, comm=shared.comm(rank), chunks=3, exchange=exchange, seed=4,
                                  fused_pack=fused, cosine_mode=("per_edge" if d == 100 else "reference"))
                assert any(m is not None for m in eng.mirrors) == (fused and exchange == "halo")
                assert eng.p2p == (exchange == "halo_p2p")
                eng.build_P()
                P_local = eng.P[:eng.E_loc].cpu()
                deltas = [eng.sweep(gamma) for _ in range(3)]
                results[rank] = (eng.get_Z(), deltas, P_local, eng.local.edge_origin, eng.exchange_bytes_per_sweep(),
                                 eng.ld)
        except Exception as exc:                                    # surface the failure, release the others
            errors.append((rank, exc))
            shared.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    P_all = torch.empty(E)
    R = world                                                       # ranks that divide the rows between them
    for Z, deltas, P_local, origin, nbytes, ld in results:
        assert O.rel_l2(Z, Z_or) < 1e-5
        for a, b in zip(deltas, deltas_or):
            assert a == pytest.approx(b, rel=1e-4)
        P_all[torch.from_numpy(origin)] = P_local
        # at most every other rank's rows (each of the 3 chunks may be padded by a row)
        assert (nbytes == 0) if exchange == "columns" else (0 < nbytes <= (R - 1) * (-(-V // R) + 3) * ld * 4)
    assert rel(P_all, P_or) < 1e-5
    if exchange in ("halo", "halo_p2p"):                            # the point of the halo: fewer bytes than all rows
        assert results[0][4] < 0.8 * (R - 1) * (V // R) * results[0][5] * 4


def test_lagged_check_on_gpu_is_bit_identical(dev):
    """lagged_check=True on the GPU (sweeps launched ahead, discarded at the stop): same sweep counts, deltas and
    embeddings, bit for bit, as the synchronous loop -- on one GPU and with 3 column-split ranks as threads."""
    csr = synth.rmat_csr(20_000, 200_000, seed=7)
    X = synth.gaussian_X(20_000, 64, seed=8)
    runs = []
    for lagged in (False, True):
        g = Graph.from_csr(csr, X)
        emb = Embedder(g, CosineSimilarity(), dev, tolerence=3, verbose=False, lagged_check=lagged, max_sweeps=300)
        emb.iterate()
        runs.append((emb.sweep_counts, emb.outer_deltas, g.Z, emb.sweeps_launched, g._engine.sweeps_done))
    a, b = runs
    assert a[0] == b[0] and a[1] == b[1] and torch.equal(a[2], b[2]) and a[3] == b[3] == a[4] == b[4]
    assert len(a[0]) >= 3 and sum(a[0]) > 20


def test_graph_without_edges(dev):
    """No edge at all (only reachable through Graph.from_csr: the reference's loader rejects an empty E file):
    nothing ever changes, every delta is 0, the countdowns run out, Z stays X."""
    csr = HostCSR(5, np.zeros(6, dtype=np.int64), np.zeros(0, dtype=np.int32))
    X = synth.gaussian_X(5, 8, seed=1)
    g = Graph.from_csr(csr, X)
    assert g.build_P(CosineSimilarity()).values().numel() == 0
    emb = Embedder(g, CosineSimilarity(), dev, tolerence=3, verbose=False)
    emb.iterate()
    assert torch.equal(g.Z, X) and emb.outer_deltas[-1] == 0.0 and emb.sweep_counts[0] == 4


def test_edge_cases_single_vertex_and_nan_termination(tmp_path):
    # one vertex with a self-loop: P = [1], Z -> X / (1 - gamma)
    root = write_data_root(tmp_path / "one", ["a"], ["a"], ["a"], np.array([[1.0, -2.0, 0.5]], dtype=np.float32))
    g = Graph(root)
    assert g.build_P(CosineSimilarity()).to_dense().item() == pytest.approx(1.0)
    emb = Embedder(g, CosineSimilarity(), torch.device("cpu"), gamma=0.5, tolerence=3, verbose=False)
    emb.iterate()
    np.testing.assert_allclose(g.Z.numpy(), np.array([[2.0, -4.0, 1.0]]), rtol=1e-5)
    # all-zero content: the reference's global denominator is 0 -> NaN scores (cs(zeros, ones) is nan upstream);
    # NaN never compares smaller, so both tolerance counters run down and the loops terminate
    k = load_golden("g2_karate_csr.npz")
    root0 = write_data_root(tmp_path / "zero", k["vertex_ids"], k["edge_src"], k["edge_dst"],
                            np.zeros((34, 4), dtype=np.float32))
    g0 = Graph(root0)
    emb0 = Embedder(g0, CosineSimilarity(), torch.device("cpu"), tolerence=3, verbose=False)
    emb0.iterate()
    Z0 = g0.Z
    sink = torch.from_numpy(np.diff(g0.csr.rowptr) == 0)
    assert torch.isnan(Z0[~sink]).all() and (Z0[sink] == 0).all()
    assert emb0.sweep_counts == [3] * 3                    # 3 endures per propagate, 3 outer endures
    # the oracle does the same
    orc = O.OracleEmbedder(g0.csr.rowptr, g0.csr.colidx, torch.zeros(34, 4), tolerence=3)
    Zo = orc.iterate()
    assert torch.isnan(Zo[~sink]).all() and orc.sweep_counts == [3] * 3


@pytest.mark.parametrize("dtype,d", [(torch.float32, 256), (torch.float32, 100), (torch.float64, 64),
                                     (torch.bfloat16, 128), (torch.float32, 1433)])
def test_spmm_split_hub_rows(dev, k, dtype, d):
    """Rows cut into segments over several workgroups + in-order combine == the one-workgroup-per-row kernel."""
    csr = ragged_csr(900, seed=3, hubs=(700, 129, 64, 900, 385))
    V, acc, gamma, seg = csr.num_vertices, _hip.acc_dtype(dtype), 0.76, 128
    X = synth.gaussian_X(V, d, seed=1).to(dtype)
    Zold = (synth.gaussian_X(V, d, seed=2) * 0.5).to(dtype)
    P = O.build_P_values(csr.rowptr, csr.colidx, synth.gaussian_X(V, 8, seed=6).double()).to(acc).to(dev)
    Xd, Zo = padded(X, dtype, dev), padded(Zold, dtype, dev)
    rowptr, colidx = torch.from_numpy(csr.rowptr).to(dev), torch.from_numpy(csr.colidx).to(dev)
    deg = np.diff(csr.rowptr)
    rows = np.nonzero(deg > seg)[0].astype(np.int32)
    nseg = -(-deg[rows] // seg)
    seg_ptr = np.zeros(rows.size + 1, dtype=np.int64)
    np.cumsum(nseg, out=seg_ptr[1:])
    seg_row = np.repeat(np.arange(rows.size, dtype=np.int32), nseg)
    rows_d, seg_ptr_d, seg_row_d = (torch.from_numpy(a).to(dev) for a in (rows, seg_ptr, seg_row))
    slab = torch.full((k.spmm_split_slab_len(int(seg_ptr[-1]), d),), float("nan"), dtype=acc, device=dev)
    Zs, Zl = torch.zeros_like(Zo), torch.zeros_like(Zo)
    ps = torch.zeros(rows.size, dtype=torch.float64, device=dev)
    pl = torch.zeros(rows.size, dtype=torch.float64, device=dev)
    k.spmm_update_split(rowptr, colidx, P, rows_d, seg_ptr_d, seg_row_d, seg, 0, Zo, Xd, gamma, Zs, d, slab, ps)
    k.spmm_update_long(rowptr, colidx, P, rows_d, 16, 0, Zo, Xd, gamma, Zl, d, pl)
    Z_ref, _ = O.sweep(csr.rowptr, csr.colidx, P.cpu().double(), X.double(), Zold.double(), gamma)
    sel = torch.from_numpy(rows.astype(np.int64))
    assert rel(Zs[sel.to(dev), :d], Z_ref[sel]) < TOL[dtype]
    assert rel(Zs[sel.to(dev), :d], Zl[sel.to(dev), :d]) < (1e-13 if dtype == torch.float64 else TOL[dtype])
    assert torch.allclose(ps, pl, rtol=1e-5 if dtype != torch.bfloat16 else 2e-2)
    untouched = torch.ones(V, dtype=torch.bool)
    untouched[sel] = False
    assert float(Zs[untouched.to(dev)].abs().sum()) == 0.0           # only the listed rows are written


@pytest.mark.parametrize("dtype,d,pad", [(torch.float32, 256, True), (torch.float32, 100, True), (torch.float64, 64, True),
                                         (torch.bfloat16, 128, True), (torch.float32, 32, True), (torch.float32, 37, False),
                                         (torch.float32, 1433, True)])
@pytest.mark.parametrize("chunk", [64, 256])
def test_spmm_class_affine_rows(dev, k, dtype, d, pad, chunk):
    """clane_spmm_update_class_*: rows whose edges are sorted by (XCD class of the column, column) and cut into chunks of one
    class, chunk blocks of class b at block index 8 j + b, partial sums added per row in slot order == the oracle,
    == the one-workgroup-per-row kernel up to summation order; only the listed rows are written; two launches are
    bitwise equal; the mirror gets the finished rows."""
    from clane_amd.xcd import class_items
    from clane_amd.xcd import xcd_class
    csr0 = ragged_csr(900, seed=3, hubs=(700, 129, 64, 900, 385, 65))
    V, acc, gamma = csr0.num_vertices, _hip.acc_dtype(dtype), 0.76
    deg = np.diff(csr0.rowptr)
    rows = np.nonzero(deg > 100)[0]
    colidx = csr0.colidx.copy()
    for r in rows:                                                  # class order inside the listed rows
        a, b = csr0.rowptr[r], csr0.rowptr[r + 1]
        c = colidx[a:b]
        colidx[a:b] = c[np.lexsort((c, xcd_class(c)))]
    X = synth.gaussian_X(V, d, seed=1).to(dtype)
    Zold = (synth.gaussian_X(V, d, seed=2) * 0.5).to(dtype)
    P = torch.rand(csr0.num_edges, generator=torch.Generator().manual_seed(3)).to(acc) / 50
    Xd, Zo = padded(X, dtype, dev, None if pad else d), padded(Zold, dtype, dev, None if pad else d)
    ci_d, P_d = torch.from_numpy(colidx).to(dev), P.to(dev)
    rp_d = torch.from_numpy(csr0.rowptr).to(dev)
    items = class_items(csr0.rowptr, colidx, rows, chunk, 8)
    assert int(items["slot_ptr"][-1]) == int((items["len"] > 0).sum()) >= 8 * rows.size - 8
    on_card = class_items(csr0.rowptr, colidx, rows, chunk, 8, colidx_dev=ci_d)     # the O(E) part on the GPU: same items
    assert all(np.array_equal(items[key], on_card[key]) for key in items)
    t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    rows_d = t(rows.astype(np.int32))
    slab = torch.full((k.spmm_class_slab_len(int(items["slot_ptr"][-1]), d),), float("nan"), dtype=acc, device=dev)
    rng = np.random.default_rng(0)
    copies = np.zeros(V, dtype=np.int64)
    copies[rows] = rng.integers(1, 3, rows.size)
    row_ptr = np.zeros(V + 1, dtype=np.int64)
    np.cumsum(copies, out=row_ptr[1:])
    buf = torch.full((int(row_ptr[-1]), Zo.shape[1]), 7.0, dtype=dtype, device=dev)
    mir = _hip.Mirror(t(row_ptr), t(np.arange(int(row_ptr[-1]), dtype=np.int32)), buf)
    Zc = torch.zeros_like(Zo)
    pc = torch.zeros(rows.size, dtype=torch.float64, device=dev)
    args = (ci_d, P_d, t(items["e0"]), t(items["len"]), t(items["slot"]), 8, rows_d, t(items["slot_ptr"]), 0, Zo, Xd,
            gamma)
    k.spmm_update_class(*args, Zc, d, slab, pc, mirror=mir)
    Z_ref, _ = O.sweep(csr0.rowptr, colidx, P.double(), X.double(), Zold.double(), gamma)
    sel = torch.from_numpy(rows.astype(np.int64))
    assert rel(Zc[sel.to(dev), :d], Z_ref[sel]) < TOL[dtype]
    Zl = torch.zeros_like(Zo)
    pl = torch.zeros(rows.size, dtype=torch.float64, device=dev)
    k.spmm_update_long(rp_d, ci_d, P_d, rows_d, 16, 0, Zo, Xd, gamma, Zl, d, pl)
    assert rel(Zc[sel.to(dev), :d], Zl[sel.to(dev), :d]) < (1e-13 if dtype == torch.float64 else TOL[dtype])
    assert torch.allclose(pc, pl, rtol=1e-5 if dtype != torch.bfloat16 else 2e-2)
    want = float((Zc[sel.to(dev), :d].double() - Zo[sel.to(dev), :d].double()).abs().sum())
    assert float(pc.sum()) == pytest.approx(want, rel=1e-6)
    untouched = torch.ones(V, dtype=torch.bool)
    untouched[sel] = False
    assert float(Zc[untouched.to(dev)].abs().sum()) == 0.0           # only the listed rows are written
    if pad:
        assert float(Zc[:, d:].abs().sum()) == 0.0
    src = torch.from_numpy(np.repeat(np.arange(V), copies)).to(dev)
    assert torch.equal(buf[:, :d], Zc[src, :d])                      # every copy of every finished row
    Z2 = torch.zeros_like(Zo)
    p2 = torch.zeros_like(pc)
    k.spmm_update_class(*args, Z2, d, slab, p2)
    assert torch.equal(Z2, Zc) and torch.equal(p2, pc)               # fixed slot order: bitwise reproducible
    with pytest.raises(_hip.ClaneHipError, match="items_per_block"):
        k.spmm_update_class(ci_d, P_d, t(items["e0"]), t(items["len"]), t(items["slot"]), 2, rows_d,
                            t(items["slot_ptr"]), 0, Zo, Xd, gamma, Z2, d, slab, p2)
    with pytest.raises(ValueError, match="whole blocks"):
        k.spmm_update_class(ci_d, P_d, t(items["e0"])[:-1], t(items["len"])[:-1], t(items["slot"])[:-1], 8, rows_d,
                            t(items["slot_ptr"]), 0, Zo, Xd, gamma, Z2, d, slab, p2)


@pytest.mark.parametrize("dtype,d,pad", [(torch.float32, 256, True), (torch.float32, 100, True), (torch.float64, 64, True),
                                         (torch.bfloat16, 128, True), (torch.float32, 32, True), (torch.float32, 37, False),
                                         (torch.float32, 1433, True)])
def test_edge_score_class_affine_rows(dev, k, dtype, d, pad):
    """clane_edge_score_class_*: K1 over the class rows' work items (chunks of one XCD class per wave): raw dots,
    reference-mode and per-edge scores == the row kernels' bit for bit (the same exchange tree per edge); with the
    fused softmax every listed row == the oracle's P and == K1 raw + K2; other rows are not written."""
    from clane_amd.xcd import class_items
    from clane_amd.xcd import xcd_class
    csr0 = ragged_csr(900, seed=3, hubs=(700, 129, 64, 900, 385, 65))
    V, acc = csr0.num_vertices, _hip.acc_dtype(dtype)
    deg = np.diff(csr0.rowptr)
    rows = np.nonzero(deg > 100)[0]
    colidx = csr0.colidx.copy()
    for r in rows:
        a, b = csr0.rowptr[r], csr0.rowptr[r + 1]
        c = colidx[a:b]
        colidx[a:b] = c[np.lexsort((c, xcd_class(c)))]
    Zc = synth.gaussian_X(V, d, seed=5).to(dtype)
    Zd = padded(Zc, dtype, dev, None if pad else d)
    t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    rp_d, ci_d, rows_d = t(csr0.rowptr), t(colidx), t(rows.astype(np.int32))
    items = class_items(csr0.rowptr, colidx, rows, 64, 8)
    it = [t(items[key]) for key in ("e0", "len", "slot", "row")]
    n_slots = int(items["slot_ptr"][-1])
    sq = torch.empty(V, dtype=acc, device=dev)
    k.row_sqnorm(Zd, d, sq)
    ws = torch.zeros(k.reduce_ws_len(), dtype=torch.float64, device=dev)
    sums2 = torch.zeros(2, dtype=torch.float64, device=dev)
    k.degree_weighted_sums(sq, rp_d, t(np.bincount(colidx, minlength=V).astype(np.int32)), V, ws, sums2)
    listed = np.zeros(csr0.num_edges, dtype=bool)
    for r in rows:
        listed[csr0.rowptr[r]:csr0.rowptr[r + 1]] = True
    listed_d = t(listed)
    stats = torch.full((2 * n_slots,), float("nan"), dtype=acc, device=dev)
    for mode, s2, sqv in ((_hip.SCORE_RAW_DOT, None, None), (_hip.SCORE_REFERENCE, sums2, None),
                          (_hip.SCORE_PER_EDGE, None, sq)):
        ref = torch.zeros(csr0.num_edges, dtype=acc, device=dev)
        k.edge_score(rp_d, ci_d, V, 0, Zd, d, mode, s2, sqv, ref)                    # every row by one (sub-)wave
        got = torch.full_like(ref, 7.0)
        k.edge_score_class(rp_d, ci_d, *it, 8, rows_d, t(items["slot_ptr"]), 0, Zd, d, mode, s2, sqv, got)
        assert torch.equal(got[listed_d], ref[listed_d])
        assert bool((got[~listed_d] == 7.0).all())                                   # only the listed rows' edges
        if mode != _hip.SCORE_RAW_DOT:
            fused = torch.full_like(ref, 7.0)
            k.edge_score_class(rp_d, ci_d, *it, 8, rows_d, t(items["slot_ptr"]), 0, Zd, d, mode, s2, sqv, fused, stats,
                               fuse_softmax=True)
            k.segment_softmax(rp_d, V, ref)
            assert rel(fused[listed_d], ref[listed_d]) < (1e-14 if dtype == torch.float64 else 3e-7)
            P_or = O.build_P_values(csr0.rowptr, colidx, Zc.to(acc).double(),
                                    mode="per_edge" if mode == _hip.SCORE_PER_EDGE else "reference")
            assert rel(fused[listed_d], P_or[torch.from_numpy(listed)]) < (
                max(TOL[dtype], 1e-6) if dtype != torch.bfloat16 else 1e-4)
            for r in rows:
                assert float(fused[csr0.rowptr[r]:csr0.rowptr[r + 1]].double().sum()) == pytest.approx(1.0, abs=1e-5)
    with pytest.raises(ValueError, match="stats"):
        k.edge_score_class(rp_d, ci_d, *it, 8, rows_d, t(items["slot_ptr"]), 0, Zd, d, _hip.SCORE_PER_EDGE, None, sq,
                           got, None, fuse_softmax=True)


@pytest.mark.parametrize("dtype,d,pad", [(torch.float32, 256, True), (torch.float32, 100, True), (torch.float64, 64, True),
                                         (torch.bfloat16, 128, True), (torch.float32, 37, False),
                                         # few writer lanes, more places than lanes in the group that hands them round
                                         (torch.float32, 2, True), (torch.bfloat16, 24, True), (torch.float64, 6, True)])
def test_spmm_mirror_packs_send_buffer(dev, k, dtype, d, pad):
    """clane_mirror_t: each finished row is also stored to its slots of a second buffer (0 ... 12 slots per row: fewer
    and more than the lanes that cover a narrow row) by whichever kernel finishes it -- main pass, 4/16-wave rows,
    split hubs -- bit-identical to Z_new.  The places of a row are loaded by the group's lanes together and handed
    round lane to lane (mirror_store, csrc/spmm_update.h)."""
    csr = ragged_csr(700, seed=5, hubs=(300, 90, 650, 200, 100, 60, 77))
    V, acc, gamma, T, seg = csr.num_vertices, _hip.acc_dtype(dtype), 0.76, 48, 128
    deg = np.diff(csr.rowptr)
    X = synth.gaussian_X(V, d, seed=1).to(dtype)
    Zold = (synth.gaussian_X(V, d, seed=2) * 0.5).to(dtype)
    P = O.build_P_values(csr.rowptr, csr.colidx, synth.gaussian_X(V, 8, seed=6).double()).to(acc).to(dev)
    Xd, Zo = padded(X, dtype, dev, None if pad else d), padded(Zold, dtype, dev, None if pad else d)
    rowptr, colidx = torch.from_numpy(csr.rowptr).to(dev), torch.from_numpy(csr.colidx).to(dev)
    rng = np.random.default_rng(0)
    copies = np.where(deg > 0, rng.integers(0, 4, V), 0)          # rows without edges are never mirrored
    many = rng.random(V) < 0.1
    copies[many & (deg > 0)] = rng.integers(8, 13, int((many & (deg > 0)).sum()))
    copies[deg > T] = np.maximum(copies[deg > T], 1)              # every long row at least once
    rows_of_slot = rng.permutation(np.repeat(np.arange(V), copies))
    order = np.argsort(rows_of_slot, kind="stable").astype(np.int32)
    row_ptr = np.zeros(V + 1, dtype=np.int64)
    np.cumsum(copies, out=row_ptr[1:])
    # two destination buffers (as with peer tables): slot k lives in buffer k % 2, row k // 2
    buf = torch.full((2, (rows_of_slot.size + 1) // 2, Zo.shape[1]), 7.0, dtype=dtype, device=dev)
    place = ((order % 2) << _hip.MIRROR_ROW_BITS) | (order // 2)
    mir = _hip.Mirror(torch.from_numpy(row_ptr).to(dev), torch.from_numpy(place.astype(np.int32)).to(dev),
                      [buf[0], buf[1]])
    is_split, is_long = deg > seg, deg > T
    to_dev = lambda m: torch.from_numpy(np.nonzero(m)[0].astype(np.int32)).to(dev)  # noqa: E731
    rows_s = np.nonzero(is_split)[0]
    nseg = -(-deg[rows_s] // seg)
    seg_ptr = np.zeros(rows_s.size + 1, dtype=np.int64)
    np.cumsum(nseg, out=seg_ptr[1:])
    seg_row = torch.from_numpy(np.repeat(np.arange(rows_s.size, dtype=np.int32), nseg)).to(dev)
    slab = torch.zeros(k.spmm_split_slab_len(int(seg_ptr[-1]), d), dtype=acc, device=dev)
    partials = torch.zeros(k.spmm_partials_len(V, int(is_long.sum())), dtype=torch.float64, device=dev)
    n_main = k.spmm_partials_len(V, 0)
    Zn = Zo.clone()
    k.spmm_update_split(rowptr, colidx, P, to_dev(is_split), torch.from_numpy(seg_ptr).to(dev), seg_row, seg, 0, Zo, Xd,
                        gamma, Zn, d, slab, partials[n_main:], mirror=mir)
    mid = is_long & ~is_split
    half = np.zeros(V, dtype=bool)
    half[np.nonzero(mid)[0][::2]] = True
    off = n_main + int(is_split.sum())
    k.spmm_update_long(rowptr, colidx, P, to_dev(half), 16, 0, Zo, Xd, gamma, Zn, d, partials[off:], mirror=mir)
    k.spmm_update_long(rowptr, colidx, P, to_dev(mid & ~half), 4, 0, Zo, Xd, gamma, Zn, d,
                       partials[off + int(half.sum()):], mirror=mir)
    k.spmm_update(rowptr, colidx, P, V, 0, Zo, Xd, gamma, Zn, d, T, partials, sinks_untouched=True, mirror=mir)
    Z_ref, _ = O.sweep(csr.rowptr, csr.colidx, P.cpu().double(), X.double(), Zold.double(), gamma)
    assert rel(Zn[:, :d], Z_ref) < TOL[dtype]
    assert rows_of_slot.size > V and is_split.sum() >= 2 and mid.sum() >= 2
    k_of = torch.arange(rows_of_slot.size, device=dev)
    assert torch.equal(buf[k_of % 2, k_of // 2][:, :d], Zn[torch.from_numpy(rows_of_slot).to(dev), :d])
    # an incomplete descriptor is refused on the host
    bad = _hip.Mirror(mir.row_ptr, mir.slot, buf[0])
    bad.c.slot = None
    with pytest.raises(_hip.ClaneHipError, match="mirror"):
        k.spmm_update(rowptr, colidx, P, V, 0, Zo, Xd, gamma, Zn, d, T, partials, mirror=bad)


def test_stage_Z_overlaps_sweeps_and_keeps_each_moment(dev):
    """stage_Z(): copies taken between sweeps (more of them than staging slots, resolved late and from another
    thread) hold exactly the embeddings of their moment, although the ping-pong buffers were overwritten since."""
    import threading
    V, E, d = 50_000, 600_000, 64
    csr = synth.rmat_csr(V, E, seed=11)
    X = synth.gaussian_X(V, d, seed=12)
    eng = SweepEngine(csr, X, dev)
    eng.build_P()
    truth, staged = [], []
    for _ in range(eng.STAGE_SLOTS):                   # as many as there are slots without resolving any
        eng.sweep(0.76)
        staged.append(eng.stage_Z())
        truth.append(None)
    eng2 = SweepEngine(csr, X, dev)
    eng2.build_P()
    for i in range(eng.STAGE_SLOTS):
        eng2.sweep(0.76)
        truth[i] = eng2.get_Z()
    got = [None] * len(staged)
    resolver = threading.Thread(target=lambda: [got.__setitem__(i, st.result()) for i, st in enumerate(staged)])
    resolver.start()
    eng.sweep(0.76)
    late = eng.stage_Z()                                # waits for a slot the other thread frees
    resolver.join(timeout=60)
    for a, b in zip(got, truth):
        assert torch.equal(a, b)
    eng2.sweep(0.76)
    assert torch.equal(late.result(), eng2.get_Z()) and late.result() is late.result()


def test_custom_similarity_plugin_on_gpu(tmp_path):
    """A user-defined similarity callable (plugin protocol, reference __main__.py:39-48 / graph.py:121): called
    once with the gathered GPU batches, normalised by the HIP segmented softmax."""
    gold, g = graph_from_golden(tmp_path, "g5_symkarate_d16_g0.76.npz")
    calls = []

    def dot_sim(a, b):
        calls.append((a.is_cuda, tuple(a.shape)))
        return (a * b).sum(1)

    P = g.build_P(dot_sim)
    assert calls == [(True, (156, 16))]
    X = torch.from_numpy(gold["X"])
    ref = O.build_P_values(g.csr.rowptr, g.csr.colidx, X, similarity=lambda a, b: (a * b).sum(1))
    np.testing.assert_allclose(P.values().numpy(), ref.numpy(), rtol=1e-5, atol=1e-7)
    emb = Embedder(g, dot_sim, torch.device("cuda"), gamma=0.3, tolerence=3, verbose=False, max_sweeps=50)
    emb.propagate()
    assert torch.isfinite(g.Z).all() and (g.Z - g.X).abs().sum() > 0


def test_config2_full_iterate_vs_c_oracle(tmp_path):
    """BASELINE config 2 shape (R-MAT 200k / 4M / d=128 fp32), the WHOLE algorithm: Embedder.iterate() on the GPU
    vs the plain-C oracle driven by the same tolerance machine (embedder.py:56-108).  Sweep counts may differ by
    a few (last-ulp noise near the fixed point); the embeddings may not."""
    from oracle import clane_oracle_c as OC
    V, E, d, gamma, tol = 200_000, 4_000_000, 128, 0.76, 3
    csr = synth.rmat_csr(V, E, seed=1)
    X = synth.gaussian_X(V, d, seed=2)
    # oracle
    Z, sweeps_or, min_outer, outer_tol = X.clone(), [], float("inf"), tol
    buf = torch.empty_like(Z)
    while True:
        prev = Z.clone()
        P, _ = OC.build_P(csr.rowptr, csr.colidx, Z)
        best, left, n = float("inf"), tol, 0
        while True:
            Zn, delta = OC.sweep(csr.rowptr, csr.colidx, P, X, Z, gamma, out=buf)
            Z, buf = Zn.clone(), buf
            n += 1
            if best > delta:
                best, left = delta, tol
            else:
                left -= 1
            if left == 0 or n >= 400:
                break
        sweeps_or.append(n)
        outer = float((Z - prev).abs().sum())
        if min_outer > outer:
            min_outer, outer_tol = outer, tol
        else:
            outer_tol -= 1
        if outer_tol == 0 or len(sweeps_or) >= 40:
            break
    # product path
    g = Graph.from_csr(csr, X)
    emb = Embedder(g, CosineSimilarity(), torch.device("cuda"), gamma=gamma, tolerence=tol, verbose=False,
                   max_sweeps=400)
    emb.iterate()
    Zg = g._engine.get_Z()
    assert O.rel_l2(Zg, Z) < 1e-5                                  # north-star bar: 1e-4
    # the first propagate is far from the fixed point and well defined; everything after it (which sweep stops
    # improving, how many outer rounds) is decided by last-ulp noise in the deltas and moves with any change of
    # summation order (seen: 23 vs 41 sweeps in round 2 after a row-binning threshold changed)
    assert abs(emb.sweep_counts[0] - sweeps_or[0]) <= 8
    assert tol <= len(emb.sweep_counts) <= 40 and tol <= len(sweeps_or) <= 40


def test_bench_two_processes_on_one_gpu_match_one_process(tmp_path):
    """bench.py's N > 1 flow end to end -- torchrun, process group, input checksum, column-split engine,
    collectives on device tensors, barrier + MAX timing, one JSON line from rank 0 -- with two real processes
    sharing this box's one GPU (gloo instead of RCCL, which refuses two ranks on one device).  The delta of the
    last sweep must be the 1-process run's: same arithmetic, only the dot products of build_P are summed in
    another order."""
    import json
    import socket
    import subprocess
    import sys
    root = Path(__file__).resolve().parent.parent
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    common = ["--workload", "tiny", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-fabric-probe"]
    one = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "1"] + common, capture_output=True,
                         text=True, timeout=600, cwd=root)
    assert one.returncode == 0, one.stderr[-2000:]
    # N = 2 exactly as the driver starts it: `python bench.py --gpus 2 ...`, no launcher, WORLD_SIZE unset -- bench.py
    # starts the two ranks itself (fresh children) and relays rank 0's line
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    two = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu"]
                         + common, capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert two.returncode == 0, two.stderr[-2000:]
    assert "starting ranks" in two.stderr
    r1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    r2 = json.loads([ln for ln in two.stdout.splitlines() if ln.startswith("{")][-1])
    assert len([ln for ln in two.stdout.splitlines() if ln.startswith("{")]) == 1          # ONE JSON line
    # the first sweep of the two-rank run is checked against the C oracle too (rank 0 gathers Z)
    assert r2["parity_rel_l2_vs_oracle_after_1_sweep"] < 1e-5 and r1["parity_rel_l2_vs_oracle_after_1_sweep"] < 1e-5
    assert "cpu_baseline" not in r2                                       # N = 1 only, by the bench contract
    assert r2["n_gpus"] == 2 and "column split x2" in r2["config"]["parallelism"] and r2["scaling"] == "strong"
    assert r2["last_delta"] == pytest.approx(r1["last_delta"], rel=5e-5)
    # what a reader of an N > 1 record has to be able to check: the group the collectives really ran in ...
    comm = r2["comm"]
    assert comm["backend"] == "gloo" and comm["ranks_seen"] == 2 and len(comm["devices"]) == 2
    assert comm["shared_gpu_rehearsal"] and comm["distinct_devices"] == 1 and comm["exchange"] == "columns"
    assert {dv["rank"] for dv in comm["devices"]} == {0, 1} and len({dv["pid"] for dv in comm["devices"]}) == 2
    assert comm["exchange_bytes_per_sweep"] == 0 and comm["collective_ms_per_sweep"] > 0      # the scalar all-reduce
    assert comm["collective_detail"]["sweeps_timed"] > 0 and len(comm["ms_per_step_by_rank"]) == 2
    assert comm["ms_per_step_rank_min"] <= comm["ms_per_step_rank_max"] <= r2["ms_per_step_max"] * 1.5
    ab = comm["delta_stream_ab"]                # the delta's all-reduce on its own stream, timed beside the default
    assert "error" not in ab and ab["on_ms_per_step"] > 0 and ab["off_ms_per_step"] == pytest.approx(r2["ms_per_step"])
    # ... and north_star's literal division (rows + one all-gather per sweep) measured beside the default one
    lit = r2["north_star_literal"]
    assert lit["exchange"] == "allgather_all" and "error" not in lit and lit["value"] > 0
    assert lit["parity_rel_l2_vs_oracle_after_1_sweep"] < 1e-5 and "row split x2" in lit["parallelism"]
    assert lit["comm"]["exchange"] == "allgather_all" and lit["comm"]["exchange_bytes_per_sweep"] > 0
    assert lit["comm"]["collectives_issued"]["all_gather"] > 0 and lit["vs_main_division"] == pytest.approx(
        lit["value"] / r2["value"])
    assert lit["last_delta"] == pytest.approx(r1["last_delta"], rel=5e-5)
    assert "north_star_literal" not in r1 and "comm" not in r1
    # P itself is checked against the oracle's own build_P wherever rank 0 can see all of it
    assert r1["parity_P_rel_l2_vs_oracle"] < 2e-6 and r2["parity_P_rel_l2_vs_oracle"] < 2e-6
    # a rank that drew a different graph (test hook) is overruled by rank 0's input, not trusted and not fatal
    fixed = subprocess.run(two.args, capture_output=True, text=True, timeout=600, cwd=root,
                           env=dict(env, CLANE_BENCH_PERTURB_RANK="1"))
    assert fixed.returncode == 0 and "ranks disagree" in fixed.stderr, fixed.stderr[-2000:]
    r3 = json.loads([ln for ln in fixed.stdout.splitlines() if ln.startswith("{")][-1])
    assert r3["last_delta"] == pytest.approx(r1["last_delta"], rel=5e-5)
    for r in (r1, r2):
        assert r["steps"] == 4 and r["warmup"] == 2 and r["value"] > 0 and r["roofline"]["bound"] == "hbm"
        # SURVEY 8d's protocol: 5 blocks of `steps` sweeps, the median reported with its spread
        assert r["blocks"] == 5 and len(r["block_ms_per_step"]) == 5
        assert r["ms_per_step_min"] <= r["ms_per_step"] <= r["ms_per_step_max"]
        assert r["ms_per_step"] == pytest.approx(sorted(r["block_ms_per_step"])[2])
        assert r["value"] == pytest.approx(1e3 / r["ms_per_step"]) and 0 < r["ms_per_step_hip_events"] <= r["ms_per_step_max"]
        roof = r["roofline"]
        assert 0 < roof["frac"] <= 1.0 and roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"])
        assert roof["achieved"] <= roof["achieved_algorithmic"] + 1e-9 and roof["kernel_config"]["d"] in (64, 32)
        assert all(0 < kk["frac"] <= 1.0 for kk in roof["kernels"].values()) and roof["k3_pass"]["frac"] <= 1.0
        assert roof["traffic"] is None and "no valid PMC traffic" in roof["note"]      # nothing measured for `tiny`


def test_bench_north_star_row_partition_two_processes(tmp_path):
    """`bench.py --gpus 2 --exchange allgather_all`: north_star's literal plan -- node rows of Z divided over the
    ranks, every rank holds the full Z, one in-place all-gather of the updated rows per sweep -- as the MAIN division,
    with two real processes on this box's GPU (gloo standing in for RCCL), and the live-rows form measured behind it
    in the same run: the first sweep of each matches the C oracle, their last deltas each other's.  (The one-process
    figures and the default flow are test_bench_two_processes_on_one_gpu_match_one_process's.)"""
    import json
    import subprocess
    import sys
    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    two = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu",
                          "--exchange", "allgather_all", "--also-exchange", "allgather", "--workload", "tiny", "--steps", "4",
                          "--warmup", "2", "--blocks", "2", "--no-cpu-baseline", "--host-sync", "every-sweep",
                          "--no-delta-stream-ab", "--no-fabric-probe"], capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert two.returncode == 0, two.stderr[-2000:]
    r2 = json.loads([ln for ln in two.stdout.splitlines() if ln.startswith("{")][-1])
    assert "exchange=allgather_all over RCCL per chunk" in r2["config"]["parallelism"] and r2["n_gpus"] == 2
    assert r2["parity_rel_l2_vs_oracle_after_1_sweep"] < 1e-5 and r2["comm"]["exchange_bytes_per_sweep"] > 0
    live = r2["other_divisions"]["allgather"]
    assert "exchange=allgather over RCCL per chunk" in live["parallelism"] and "error" not in live
    assert live["parity_rel_l2_vs_oracle_after_1_sweep"] < 1e-5
    assert 0 < live["comm"]["exchange_bytes_per_sweep"] < r2["comm"]["exchange_bytes_per_sweep"]     # live rows only
    assert live["last_delta"] == pytest.approx(r2["last_delta"], rel=5e-5)


def test_bench_halo_p2p_two_processes_share_tables_through_ipc(tmp_path):
    """exchange="halo_p2p" with two real processes on this box's GPU: each maps the other's tables with hipIpc
    (clane_ipc_export / clane_ipc_open) and its kernels store finished rows straight into them."""
    import json
    import socket
    import subprocess
    import sys
    root = Path(__file__).resolve().parent.parent
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    # one timed block: the delta shrinks by gamma per sweep while Z does not, so after 5 x 6 more sweeps the two
    # divisions' different summation orders show in its 5th digit (seen: 1.37668 vs 1.37669 after 33 sweeps)
    common = ["--workload", "tiny", "--steps", "6", "--warmup", "2", "--blocks", "1", "--no-cpu-baseline",
              "--also-exchange", "none", "--no-delta-stream-ab", "--no-fabric-probe"]   # hipIpc is the point: no further divisions
    one = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "1"] + common, capture_output=True,
                         text=True, timeout=600, cwd=root)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(root / "bench.py"), "--gpus", "2",
                          "--backend", "gloo", "--share-gpu", "--exchange", "halo_p2p"] + common, capture_output=True,
                         text=True, timeout=600, cwd=root)
    assert two.returncode == 0, two.stderr[-3000:]
    r1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    r2 = json.loads([ln for ln in two.stdout.splitlines() if ln.startswith("{")][-1])
    assert "hipIpc" in r2["config"]["parallelism"]
    assert r2["last_delta"] == pytest.approx(r1["last_delta"], rel=1e-5)


def test_cli_two_processes_on_one_gpu_match_one_process(tmp_path):
    """`torchrun ... -m clane_amd` with two real processes (gloo, both on this box's GPU): per-rank engine on its
    column slice, broadcast content, collective decisions, rank 0 writes -- the same Z.npy as one process."""
    import socket
    import subprocess
    import sys
    root = Path(__file__).resolve().parent.parent
    kc = load_golden("g2_karate_csr.npz")
    Xc = np.random.default_rng(5).standard_normal((34, 64)).astype(np.float32)
    data = write_data_root(tmp_path / "karate64", kc["vertex_ids"], kc["edge_src"], kc["edge_dst"], Xc)
    cfg = tmp_path / "config.yaml"
    cfg.write_text("graph:\n  embedding_dim: 64\n\nsimilarity:\n  method: \"CosineSimilarity\"\n  kwargs: {}\n\n"
                   "embedder:\n  gamma: 0.76\n  tolerence: 5\n")
    common = ["--data_root", str(data), "--config_file", str(cfg)]
    one = subprocess.run([sys.executable, "-m", "clane_amd"] + common + ["--output_root", str(tmp_path / "o1")],
                         capture_output=True, text=True, timeout=600, cwd=root)
    assert one.returncode == 0, one.stderr[-2000:]
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, CLANE_DIST_BACKEND="gloo", CLANE_SHARE_GPU="1")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "clane_amd"] + common +
                         ["--output_root", str(tmp_path / "o2")], capture_output=True, text=True, timeout=600, cwd=root,
                         env=env)
    assert two.returncode == 0, two.stderr[-2000:]
    Z1, Z2 = np.load(tmp_path / "o1" / "Z.npy"), np.load(tmp_path / "o2" / "Z.npy")
    assert Z1.shape == (34, 64) and np.linalg.norm(Z1 - Z2) <= 1e-5 * np.linalg.norm(Z1)
    assert two.stdout.count("Graph Loaded.") == 1                      # rank 0 talks, the other rank is silent
    # --exchange (extension): north_star's row partition + all-gather per sweep gives the same embeddings
    rows = subprocess.run(two.args[:-1] + [str(tmp_path / "o3"), "--exchange", "allgather_all"], capture_output=True,
                          text=True, timeout=600, cwd=root, env=env)
    assert rows.returncode == 0, rows.stderr[-2000:]
    Z3 = np.load(tmp_path / "o3" / "Z.npy")
    assert np.linalg.norm(Z1 - Z3) <= 1e-5 * np.linalg.norm(Z1)
    # --save_history on several GPUs: every rank stages its own part of Z without a collective, rank 0's
    # writer thread puts the parts together -- the same files as one process writes, for a column and for a row split
    h1 = subprocess.run(one.args[:-1] + [str(tmp_path / "h1"), "--save_history"], capture_output=True, text=True,
                        timeout=600, cwd=root)
    assert h1.returncode == 0, h1.stderr[-2000:]
    files1 = sorted(str(p.relative_to(tmp_path / "h1")) for p in (tmp_path / "h1").rglob("Z_*.npy"))
    assert len(files1) > 10
    for tag, more in (("h2", []), ("h3", ["--exchange", "allgather_all"])):
        hn = subprocess.run(two.args[:-1] + [str(tmp_path / tag), "--save_history"] + more, capture_output=True,
                            text=True, timeout=600, cwd=root, env=env)
        assert hn.returncode == 0, hn.stderr[-2000:]
        out = tmp_path / tag
        assert not (out / ".clane_history_parts").exists()                  # every part was consumed
        # sweep counts near the fixed point sit on last-ulp noise: compare the sweeps both runs have
        files_n = sorted(str(p.relative_to(out)) for p in out.rglob("Z_*.npy"))
        both = sorted(set(files1) & set(files_n))
        assert len(both) >= 0.8 * len(files1) and "0/Z_0.npy" in both
        for rel_path in both:
            a, b = np.load(tmp_path / "h1" / rel_path), np.load(out / rel_path)
            assert a.shape == b.shape == (34, 64) and np.linalg.norm(a - b) <= 1e-5 * np.linalg.norm(a), rel_path
        assert np.linalg.norm(np.load(out / "Z.npy") - Z1) <= 1e-5 * np.linalg.norm(Z1)


def test_config3_full_size_properties(dev):
    """BASELINE config 3 at full size (R-MAT 2M / 40M / d=256 fp32): (0) all of P and the whole first sweep against the
    C oracle running its own build_P and sweep (seconds on the box's host cores); and what does not need the
    oracle: (1) every row of P with edges sums to 1 (graph.py:122-123); (2) 2 000 sampled rows of the first sweep
    against a float64 restatement of embedder.py:88-92 on those rows; (3) rows without out-edges keep z;
    (4) the reported delta is sum|Z_new - Z_old|; (5) two engines give bit-identical results; (6) with P frozen
    the sweep is affine in Z:  F(Za) - F(Zb) = F(Za - Zb) - X."""
    V, E, d, gamma = 2_000_000, 40_000_000, 256, 0.76
    csr = synth.rmat_csr(V, E, seed=3, device=str(dev))
    X = synth.gaussian_X(V, d, seed=4)
    eng = SweepEngine(csr, X, dev)
    eng.build_P()
    P = eng.P_global()                                         # CPU, global (row, col) order
    deg = np.diff(csr.rowptr)
    cs = torch.cat([torch.zeros(1, dtype=torch.float64), P.double().cumsum(0)])
    sums = (cs[torch.from_numpy(csr.rowptr[1:])] - cs[torch.from_numpy(csr.rowptr[:-1])]).numpy()
    assert np.abs(sums[deg > 0] - 1).max() < 1e-4 and np.abs(sums[deg == 0]).max() == 0
    # (0) reference-mode P itself, all 40M values, against the C oracle's OWN build_P (graph.py:118-128 +
    # similarity.py:35-37 restated in oracle/clane_oracle.c) -- nothing of the GPU's goes into the expected side
    P_or, D_or = OC.build_P(csr.rowptr, csr.colidx, X)
    assert P_or.numel() == P.numel() == E
    assert O.rel_l2(P, P_or) < 2e-6 and float((P - P_or).abs().max()) < 1e-6
    hubs = np.argsort(deg)[-3:]
    for r in hubs:                                             # the heaviest rows on their own (class pass + combine)
        a, b = csr.rowptr[r], csr.rowptr[r + 1]
        assert O.rel_l2(P[a:b], P_or[a:b]) < 2e-6, r
    delta = eng.sweep(gamma)
    Z1 = eng.get_Z()
    # the whole first sweep against the oracle's sweep run with the ORACLE's P
    Z1_or, delta_or = OC.sweep(csr.rowptr, csr.colidx, P_or, X, X, gamma)
    assert O.rel_l2(Z1, Z1_or) < 1e-6 and delta == pytest.approx(delta_or, rel=1e-6)
    del Z1_or
    rows = np.random.default_rng(0).choice(V, size=2000, replace=False)
    rows = np.concatenate([rows, np.argsort(deg)[-3:]])        # and the three heaviest hubs
    Pn, Xn = P.numpy().astype(np.float64), X.numpy()
    for r in rows:
        a, b = csr.rowptr[r], csr.rowptr[r + 1]
        want = Xn[r].astype(np.float64) if a == b else \
            Xn[r] + gamma * (Pn[a:b, None] * Xn[csr.colidx[a:b]].astype(np.float64)).sum(0)
        got = Z1[r].numpy().astype(np.float64)
        assert np.linalg.norm(got - want) <= 2e-6 * max(np.linalg.norm(want), 1e-30), r
    sink = torch.from_numpy(deg == 0)
    assert torch.equal(Z1[sink], X[sink])
    assert delta == pytest.approx(float((Z1.double() - X.double()).abs().sum()), rel=1e-6)
    eng_b = SweepEngine(csr, X, dev)
    eng_b.build_P()
    assert torch.equal(eng_b.P, eng.P) and eng_b.sweep(gamma) == delta and torch.equal(eng_b.Zcur, eng.Zcur)
    del eng_b
    # affine in Z with P frozen (set_Z drops P: put the flag back, P itself is untouched)
    g = torch.Generator().manual_seed(1)
    Za = torch.randn(V, d, generator=g)
    Zb = torch.randn(V, d, generator=g)

    def F(Z):
        eng.set_Z(Z)
        eng.P_valid = True
        eng.sweep(gamma)
        return eng.get_Z()
    lhs = F(Za) - F(Zb)
    rhs = F(Za - Zb) - X
    live = ~sink                                               # rows without out-edges return z itself: F(z) = z
    assert O.rel_l2(lhs[live], rhs[live]) < 1e-5
    assert torch.equal(lhs[sink], (Za - Zb)[sink])


def test_integration_stub_from_the_docs_runs(tmp_path):
    """INTEGRATION.md section B, executed as written (only the library path is made absolute): the ctypes stub a
    reference maintainer would add drives one sweep through the C ABI and matches the oracle."""
    import re
    root = Path(__file__).resolve().parent.parent
    text = (root / "INTEGRATION.md").read_text()
    code = re.search(r"```python\n(# clane/_hip_backend.py.*?)```", text, flags=re.S).group(1)
    code = code.replace('C.CDLL("libclane_hip.so")', f'C.CDLL("{_hip.LIB_PATH}")')
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    gold, g = graph_from_golden(tmp_path, "g5_symkarate_d16_g0.76.npz")
    state = ns["HipState"](g)
    P_or = O.build_P_values(g.csr.rowptr, g.csr.colidx, g.X)
    state.P.copy_(P_or.float())
    delta = state.sweep(0.76)
    Z_or, d_or = O.sweep(g.csr.rowptr, g.csr.colidx, P_or, g.X, g.X, 0.76)
    assert O.rel_l2(state.Z[state.cur].cpu(), Z_or) < 1e-6 and delta == pytest.approx(float(d_or), rel=1e-5)


def test_lane_xor_exchanges_on_the_card(tmp_path):
    """The DPP / v_permlane*_swap forms of lane ^ M (device_utils.h: every butterfly of K1 / K3 stands on them) against
    __shfl_xor, compiled here with the box's hipcc and run on the card: a compiler or ROCm bump that changes what these
    builtins do is caught by the suite, not by a wrong delta."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = tmp_path / "lane_xor_check"
    src = Path(__file__).resolve().parent.parent / "tools" / "lane_xor_check.hip"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++20", f"-I{src.parent.parent / 'clane_amd' / 'csrc'}",
                    "-o", str(exe), str(src)], check=True, timeout=300)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "lane_xor check: 0 mismatches" in out.stdout, out.stdout + out.stderr


def test_tables_in_contiguous_allocations(dev):
    """SweepEngine(table_alloc="contiguous"): the Z tables and X in physically contiguous allocations of their own
    (clane_device_alloc_contiguous) -- an experiment's knob (profiles/r04_placement_probe_box4_contiguous.jsonl), but a
    product option: same results bit for bit, the allocations live as long as the engine, an unknown choice is refused."""
    csr = synth.rmat_csr(20_000, 200_000, seed=7)
    X = synth.gaussian_X(20_000, 64, seed=8)
    a, b = SweepEngine(csr, X, dev), SweepEngine(csr, X, dev, table_alloc="contiguous")
    assert len(b._own_tables) == 2 + 1 or b.table_alloc_note is not None   # two Z tables + X; or the driver had no such range
    for eng in (a, b):
        eng.build_P()
        for _ in range(3):
            eng.sweep(0.76)
    assert torch.equal(a.get_Z(), b.get_Z()) and torch.equal(a.P, b.P)
    b.snapshot()                                # the third Z table comes with the first snapshot, from the same allocator
    assert len(b._own_tables) == SweepEngine.N_TABLES + 1 or b.table_alloc_note is not None
    for eng in (a, b):
        eng.sweep(0.76)
    assert torch.equal(a.get_Z(), b.get_Z())
    with pytest.raises(ValueError, match="table_alloc"):
        SweepEngine(csr, X, dev, table_alloc="pinned")
