"""GPU parity at scale, in the modes where the edge SCORES matter.

In the reference's own mode (similarity.py:37: one global denominator) a score is about dot / (|E| d) ~ 1e-8 and the
row softmax equals 1/deg to fp32 epsilon, so a P that matches the oracle says nothing about the dot products behind
it.  The tests here therefore use the raw dot products (CLANE_SCORE_RAW_DOT) and the true per-edge
cosine (`cosine_mode="per_edge"`), whose softmax moves with every score, on graphs whose hub rows are walked in
SEVERAL 64-edge chunks per wave of `edge_score_long_kernel` (rows above 1024 edges: running max / sum per wave, the
LDS {max, sum} combine, the rescale of the wave's own stores) -- config 2 and config 3 sizes, d = 128 and 256,
fp32 / fp64 / bf16.  Config 4 (power-law 10M / 200M / d=128 bf16) runs at its full shape on one GPU.

Tolerances as in test_gpu_parity.py: fp32 <= 2e-6 rel-L2 (summation order), fp64 <= 1e-13, bf16 storage <= 8e-3.
"""
import importlib.util
import socket
from pathlib import Path

import numpy as np
import pytest
import torch

from clane_amd import _hip, synth
from clane_amd.embedder import Embedder
from clane_amd.engine import SweepEngine, lanes_per_row
from clane_amd.graph import Graph
from clane_amd.partition import HostCSR
from clane_amd.similarity import CosineSimilarity
from oracle import clane_oracle as O

from .conftest import load_golden, write_data_root
from .test_gpu_parity import TOL, padded, rel

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def dev():
    return _hip.require_gpu("cuda:0")


@pytest.fixture(scope="module")
def k():
    return _hip.kernels()


def hub_csr(V, hubs, seed, max_deg=6):
    """Sparse background (0..max_deg edges per row, some empty) plus rows of exactly the given degrees."""
    rng = np.random.default_rng(seed)
    deg = rng.integers(0, max_deg + 1, size=V)
    where = rng.choice(V, size=len(hubs), replace=False)
    deg[where] = hubs
    rowptr = np.zeros(V + 1, dtype=np.int64)
    np.cumsum(deg, out=rowptr[1:])
    cols = np.empty(int(rowptr[-1]), dtype=np.int32)
    small = rng.integers(0, V, size=(V, max_deg + 2))
    for r in range(V):
        dg = deg[r]
        if dg == 0:
            continue
        if dg <= max_deg:
            c = np.unique(small[r])[:dg]
            while c.size < dg:                       # the few rows whose draws collided
                c = np.unique(np.concatenate([c, rng.integers(0, V, size=dg)]))[:dg]
        else:
            c = np.sort(rng.choice(V, size=dg, replace=False))
        cols[rowptr[r]:rowptr[r + 1]] = c
    return HostCSR(V, rowptr, cols), where


def per_edge_rows_f64(csr, Z, rows):
    """float64 restatement, on the given rows only, of per-edge cosine + row softmax (similarity.py:26-37 with
    per-pair norms, graph.py:122-123): {row: P values of the row}."""
    out = {}
    Zn = Z.numpy() if isinstance(Z, torch.Tensor) else Z
    for r in rows:
        a, b = int(csr.rowptr[r]), int(csr.rowptr[r + 1])
        if a == b:
            out[int(r)] = np.empty(0)
            continue
        zs = Zn[r].astype(np.float64)
        nb = Zn[csr.colidx[a:b]].astype(np.float64)
        s = (nb @ zs) / (np.linalg.norm(zs) * np.linalg.norm(nb, axis=1))
        e = np.exp(s - s.max())
        out[int(r)] = e / e.sum()
    return out


# ---- (c) hubs of 1 100 / 5 000 / 20 000 edges: one wave of the 16-wave row kernel walks >= 2 chunks -----------------
@pytest.mark.parametrize("dtype,d", [(torch.float32, 256), (torch.float64, 256), (torch.bfloat16, 256),
                                     (torch.bfloat16, 128), (torch.float32, 128), (torch.float32, 100)])
def test_k1_multi_chunk_hub_rows(dev, k, dtype, d):
    """Rows of 1 025 ... 20 000 edges through edge_score_long_kernel (16 waves: 2 ... 20 chunks of 64 edges per
    wave): raw dots and per-edge P against the oracle, fused softmax == K1 raw + K1b finalize + K2 (the
    column-split route), sliced scores == the one-wave kernel's bit for bit."""
    V = 20_500
    csr, where = hub_csr(V, [1025, 1100, 5000, 20000, 1024, 64, 65, 2047], seed=d)
    acc = _hip.acc_dtype(dtype)
    Zc = synth.gaussian_X(V, d, seed=11).to(dtype)
    Zf = Zc.to(acc).double()                                            # what the kernels see after widening
    Zd = padded(Zc, dtype, dev)
    rowptr, colidx = torch.from_numpy(csr.rowptr).to(dev), torch.from_numpy(csr.colidx).to(dev)
    deg = np.diff(csr.rowptr)
    T = 32 if lanes_per_row(d, dtype) == 64 else 128                  # the engine's defaults for these row widths
    long_rows = torch.from_numpy(np.nonzero(deg > T)[0].astype(np.int32)).to(dev)
    assert long_rows.numel() == int((np.array([1025, 1100, 5000, 20000, 1024, 64, 65, 2047]) > T).sum()) >= 6
    tol = 5e-6 if dtype == torch.bfloat16 else TOL[dtype]

    dots_ref = O.edge_dots(csr.rowptr, csr.colidx, Zf)
    raw = torch.full((csr.num_edges,), float("nan"), dtype=acc, device=dev)
    k.edge_score(rowptr, colidx, V, 0, Zd, d, _hip.SCORE_RAW_DOT, None, None, raw, T, long_rows)
    assert rel(raw, dots_ref) < tol
    whole = torch.full_like(raw, float("nan"))                          # every row by one (sub-)wave
    k.edge_score(rowptr, colidx, V, 0, Zd, d, _hip.SCORE_RAW_DOT, None, None, whole)
    assert torch.equal(whole, raw)
    for h in where[:4]:                                                 # the hub rows on their own
        a, b = csr.rowptr[h], csr.rowptr[h + 1]
        assert rel(raw[a:b], dots_ref[a:b]) < tol, (h, b - a)

    sq = torch.empty(V, dtype=acc, device=dev)
    k.row_sqnorm(Zd, d, sq)
    fused = torch.full_like(raw, float("nan"))
    k.edge_score(rowptr, colidx, V, 0, Zd, d, _hip.SCORE_PER_EDGE, None, sq, fused, T, long_rows, fuse_softmax=True)
    P_ref = O.build_P_values(csr.rowptr, csr.colidx, Zf, mode="per_edge")
    assert rel(fused, P_ref) < max(tol, 1e-6)
    for h in where[:4]:
        a, b = csr.rowptr[h], csr.rowptr[h + 1]
        assert rel(fused[a:b], P_ref[a:b]) < max(tol, 1e-6), (h, b - a)
        assert float(fused[a:b].double().sum()) == pytest.approx(1.0, abs=1e-5)
    # the non-fused route of the column split: raw dots, denominators applied afterwards, K2 over every row
    late = raw.clone()
    k.edge_score_finalize(rowptr, colidx, V, 0, _hip.SCORE_PER_EDGE, None, sq, late)
    k.segment_softmax(rowptr, V, late, 0, T, long_rows)
    assert rel(late, fused) < (1e-14 if dtype == torch.float64 else 3e-7)
    # reference mode through the same long rows (global denominator): softmax ~ 1/deg, and it must still be exact
    ws = torch.zeros(k.reduce_ws_len(), dtype=torch.float64, device=dev)
    sums2 = torch.zeros(2, dtype=torch.float64, device=dev)
    k.degree_weighted_sums(sq, rowptr, torch.from_numpy(csr.indeg()).to(dev), V, ws, sums2)
    k.edge_score(rowptr, colidx, V, 0, Zd, d, _hip.SCORE_REFERENCE, sums2, None, fused, T, long_rows, fuse_softmax=True)
    assert rel(fused, O.build_P_values(csr.rowptr, csr.colidx, Zf)) < max(tol, 1e-6)


# ---- (a) config 2 size, default thresholds, d = 128 (2 rows per wave) and d = 256 (a row fills a wave) -------------
@pytest.mark.parametrize("d", [128, 256])
def test_k1_scores_at_scale(dev, k, d):
    """R-MAT 200k / 4M: `SweepEngine(cosine_mode="per_edge").build_P()` (the engine's own row order, thresholds and
    long-row lists) against O.build_P_values(mode="per_edge"); raw dot products of all 4M edges against
    O.edge_dots at <= 2e-6; the rows above 1 024 edges checked on their own."""
    V, E = 200_000, 4_000_000
    csr = synth.rmat_csr(V, E, seed=1)
    deg = np.diff(csr.rowptr)
    big = np.nonzero(deg > 1024)[0]
    assert big.size >= 20                                               # multi-chunk rows exist at this size
    X = synth.gaussian_X(V, d, seed=2)
    eng = SweepEngine(csr, X, dev, cosine_mode="per_edge")
    assert eng.score_threshold == (64 if d == 256 else 128) and eng.class_k1 and eng.class_rows[0] is not None
    eng.build_P()
    P_gpu = eng.P_global()
    P_ref = O.build_P_values(csr.rowptr, csr.colidx, X, mode="per_edge")
    assert rel(P_gpu, P_ref) < 2e-6
    worst = max(rel(P_gpu[csr.rowptr[r]:csr.rowptr[r + 1]], P_ref[csr.rowptr[r]:csr.rowptr[r + 1]]) for r in big)
    assert worst < 5e-6, worst
    # raw dots with the engine's own structure (its vertex order, its long-row list)
    raw = torch.full((eng.E_loc,), float("nan"), dtype=torch.float32, device=dev)
    k.edge_score(eng.rowptr, eng.colidx, eng.part.n_local, 0, eng.Zcur, eng.d, _hip.SCORE_RAW_DOT, None, None, raw,
                 eng.k1_threshold, eng.k1_long_rows[0])
    rows_c, slot_ptr, it_e0, it_len, it_slot, it_row, ipb = eng.class_rows[0]    # ... and its class rows' work items
    k.edge_score_class(eng.rowptr, eng.colidx, it_e0, it_len, it_slot, it_row, ipb, rows_c, slot_ptr, 0, eng.Zcur, eng.d,
                       _hip.SCORE_RAW_DOT, None, None, raw)
    dots = torch.empty(E)
    dots[torch.from_numpy(eng.local.edge_origin)] = raw.cpu()
    dots_ref = O.edge_dots(csr.rowptr, csr.colidx, X)
    assert rel(dots, dots_ref) < 2e-6
    worst = max(rel(dots[csr.rowptr[r]:csr.rowptr[r + 1]], dots_ref[csr.rowptr[r]:csr.rowptr[r + 1]]) for r in big)
    assert worst < 2e-6, worst
    # and the reference mode on the same engine object (what bench.py and the CLI run)
    eng.cosine_mode, eng.P_valid = "reference", False
    eng.build_P()
    assert rel(eng.P_global(), O.build_P_values(csr.rowptr, csr.colidx, X)) < 2e-6
    # the long-row kernels of round 1 (no class pass anywhere) give the same per-edge P
    old = SweepEngine(csr, X, dev, cosine_mode="per_edge", class_threshold=0)
    assert old.long_rows[0] is not None and not old.class_k1
    old.build_P()
    assert rel(old.P_global(), P_ref) < 2e-6 and rel(old.P_global(), P_gpu) < 1e-6


# ---- (b) config 3 at full size: per-edge P on sampled rows + the heaviest hubs -----------------------------------------
def test_config3_per_edge_P_sampled_rows(dev):
    """R-MAT 2M / 40M / d=256 fp32 (188k rows above 32 edges, heaviest out-degree ~70k): per-edge P of 2 000 sampled
    rows and the three heaviest hubs against a float64 restatement of similarity.py:26-37 (per-pair norms) and
    graph.py:122-123 on those rows; every row with edges sums to 1."""
    V, E, d = 2_000_000, 40_000_000, 256
    csr = synth.rmat_csr(V, E, seed=3, device=str(dev))
    X = synth.gaussian_X(V, d, seed=4)
    eng = SweepEngine(csr, X, dev, cosine_mode="per_edge")
    eng.build_P()
    P = eng.P_global().numpy()
    deg = np.diff(csr.rowptr)
    rng = np.random.default_rng(1)
    rows = np.concatenate([rng.choice(V, size=1500, replace=False),
                           rng.choice(np.nonzero(deg > 1024)[0], size=500, replace=False), np.argsort(deg)[-3:]])
    want = per_edge_rows_f64(csr, X, rows)
    for r in rows:
        a, b = csr.rowptr[r], csr.rowptr[r + 1]
        if b > a:
            got = P[a:b].astype(np.float64)
            assert np.linalg.norm(got - want[int(r)]) <= 3e-6 * np.linalg.norm(want[int(r)]), (r, b - a)
    cs = np.concatenate([[0.0], np.cumsum(P.astype(np.float64))])
    sums = cs[csr.rowptr[1:]] - cs[csr.rowptr[:-1]]
    assert np.abs(sums[deg > 0] - 1).max() < 1e-4 and np.abs(sums[deg == 0]).max() == 0
    # unlike the reference mode, this P is far from uniform: the check above can tell a wrong dot product
    hub = int(np.argmax(deg))
    ph = P[csr.rowptr[hub]:csr.rowptr[hub + 1]]
    assert ph.max() / ph.min() > 1.2


# ---- (d) config 4 at its full shape on one GPU -------------------------------------------------------------------------
def test_config4_full_shape(dev):
    """BASELINE config 4: power-law |V| = 10M, |E| ~ 200M, d = 128, bf16 storage (fp32 accumulate, fp32 P), one GPU.
    (0) ALL 200M values of P and the WHOLE first sweep against oracle/clane_oracle.c running its own build_P
    (graph.py:118-128 + similarity.py:26-37) and sweep (embedder.py:84-94) in fp32 on the bf16-rounded X: P within 2e-6
    rel-L2 (fp32 arithmetic on both sides), Z1 within 8e-3 (one bf16 rounding of every stored value) -- the test fails
    if the bf16 K1 (edge_score_subrow) or K3 (spmm_update_subrow) kernels drift from the oracle at full size;
    (1) P rows sum to 1; (2) 2 000 sampled rows + the three heaviest hubs of the first sweep against fp32/float64
    arithmetic on the bf16-rounded inputs (embedder.py:88-92; <= 8e-3: one bf16 rounding of the result) -- a property
    check with the GPU's own P; (3) rows without out-edges keep z; (4) delta = sum|Z_new - Z_old|;
    (5) Embedder.iterate() from Z = X runs to `tolerence` with the outer delta reaching 0 (the bf16 fixed point);
    (6) at that point the sampled rows satisfy z = x + gamma * P z to bf16 rounding with the P of the final embeddings
    (graph.py:118-128 + embedder.py:92).  NOTE on (0): at this size the reference-mode scores are O(1e-9) (global
    Frobenius denominators, similarity.py:37), exp() of them is 1 within fp32 and P = 1/deg on both sides -- the P
    comparison checks the softmax plumbing and the edge order, not K1's dot products; those are checked at scale by
    test_k1_scores_at_scale and test_config3_per_edge_P_sampled_rows (raw dots / per-edge cosine)."""
    from oracle import clane_oracle_c as OC
    V, E, d, gamma = 10_000_000, 200_000_000, 128, 0.76
    csr = synth.powerlaw_csr(V, E, seed=5, device=str(dev))
    assert csr.num_edges == E                   # exactly BASELINE's 200M distinct edges (rounds 1-3: 198M)
    X = synth.gaussian_X(V, d, seed=6).to(torch.bfloat16)
    g = Graph.from_csr(csr, X)
    eng = g.engine(dev)
    assert eng.dtype == torch.bfloat16 and eng.P.dtype == torch.float32
    eng.build_P()
    P = eng.P_global().numpy()
    deg = np.diff(csr.rowptr)
    cs = np.concatenate([[0.0], np.cumsum(P.astype(np.float64))])
    sums = cs[csr.rowptr[1:]] - cs[csr.rowptr[:-1]]
    assert np.abs(sums[deg > 0] - 1).max() < 1e-4
    del cs, sums
    delta = eng.sweep(gamma)
    Z1 = eng.get_Z()
    # (0) the C oracle's own P and first sweep, all of both
    Xf = X.float()
    P_c, _ = OC.build_P(csr.rowptr, csr.colidx, Xf)
    assert O.rel_l2(torch.from_numpy(P), P_c) < 2e-6
    Z1_c, delta_c = OC.sweep(csr.rowptr, csr.colidx, P_c, Xf, Xf, gamma)
    assert O.rel_l2(Z1.float(), Z1_c) < 8e-3
    assert delta == pytest.approx(delta_c, rel=2e-2)        # sum of |differences| of bf16-rounded values
    del Xf, P_c, Z1_c
    rng = np.random.default_rng(0)
    rows = np.concatenate([rng.choice(V, size=2000, replace=False), np.argsort(deg)[-3:]])

    def residual(Z_in, Z_out, Pv):
        worst = 0.0
        for r in rows:
            a, b = csr.rowptr[r], csr.rowptr[r + 1]
            if a == b:
                assert torch.equal(Z_out[r], X[r])
                continue
            nb = Z_in[torch.from_numpy(csr.colidx[a:b].astype(np.int64))].double().numpy()
            want = X[r].double().numpy() + gamma * (Pv[a:b, None].astype(np.float64) * nb).sum(0)
            got = Z_out[r].double().numpy()
            worst = max(worst, np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30))
        return worst
    assert residual(X, Z1, P) < 8e-3
    sink = torch.from_numpy(deg == 0)
    if bool(sink.any()):
        assert torch.equal(Z1[sink], X[sink])
    assert delta == pytest.approx(float((Z1.float() - X.float()).abs().sum(dtype=torch.float64)), rel=1e-5)
    del Z1

    eng.set_Z(X)
    emb = Embedder(g, CosineSimilarity(), dev, gamma=gamma, tolerence=10, verbose=False, max_sweeps=2000)
    emb.iterate()
    assert emb.tolerences["global"].value == 0 and emb.outer_deltas[-1] == 0.0
    assert 10 <= len(emb.sweep_counts) <= 40 and emb.sweep_counts[0] > 20
    assert emb.outer_deltas[0] > 1e6 * max(emb.outer_deltas[1], 1e-30) or emb.outer_deltas[1] == 0.0
    Zf = eng.get_Z()
    assert torch.isfinite(Zf.float()).all()
    eng.build_P()
    assert residual(Zf, Zf, eng.P_global().numpy()) < 8e-3           # the fixed point of the reference's update


# ---- RCCL on the hardware that exists: a one-rank "nccl" group on cuda:0 -----------------------------------------------
def test_rccl_single_rank_group_drives_the_engine_collectives(dev):
    """backend="nccl" IS RCCL on ROCm.  One rank cannot show scaling, but it runs every collective the engine issues
    through the real library on device tensors: the in-place all-gather and the split-size all-to-all with
    `async_op=True` issued from side streams, the float64 scalar all-reduce, and a whole row-partitioned sweep
    (`exchange="allgather_all"`: north_star's row partition + in-place all-gather) whose result must be the plain
    one-GPU engine's, bit for bit."""
    import torch.distributed as dist
    from clane_amd.comm import TorchComm
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    assert not dist.is_initialized()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        pg = dist.group.WORLD
        comm = TorchComm(pg, force_collectives=True)
        t = torch.arange(5, dtype=torch.float64, device=dev)
        comm.all_reduce_sum(t)
        assert torch.equal(t.cpu(), torch.arange(5, dtype=torch.float64))
        for dtype in (torch.float32, torch.bfloat16, torch.float64):
            table = torch.zeros(64, 32, dtype=dtype, device=dev)
            mine = table[16:48]                                         # in-place form: the send buffer is a slice of
            mine.copy_(torch.randn(32, 32, device=dev).to(dtype))       # the receive buffer
            keep = mine.clone()
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                w = comm.all_gather_into(table[16:48], mine, async_op=True)
            w.wait()
            assert torch.equal(table[16:48], keep) and float(table[:16].abs().sum()) == 0
            send = torch.randn(40, 32, device=dev).to(dtype)
            recv = torch.zeros(40, 32, dtype=dtype, device=dev)
            with torch.cuda.stream(side):
                w = comm.all_to_all_rows(recv, send, [40], [40], async_op=True)
            w.wait()
            torch.cuda.current_stream(dev).synchronize()
            assert torch.equal(recv, send)
        # a whole engine over the group: chunked blocks, RCCL all-gather per chunk on side streams, scalar all-reduce
        V, E, d, gamma = 20_000, 300_000, 64, 0.76
        csr = synth.rmat_csr(V, E, seed=9)
        X = synth.gaussian_X(V, d, seed=10)
        plain = SweepEngine(csr, X, dev)
        plain.build_P()
        for exchange in ("allgather_all", "allgather", "columns"):
            eng = SweepEngine(csr, X, dev, comm=TorchComm(pg, force_collectives=True), exchange=exchange, chunks=3,
                              shuffle=False)
            assert eng.exchange == exchange and eng.comm.force
            eng.build_P()
            assert rel(eng.P_global(), plain.P_global()) < 2e-6
            plain.set_Z(X)
            plain.P_valid = True
            for _ in range(3):
                da, db = eng.sweep(gamma), plain.sweep(gamma)
                assert da == pytest.approx(db, rel=1e-6)
            assert O.rel_l2(eng.get_Z(), plain.get_Z()) < 2e-6
            assert eng.comm.calls["all_reduce"] >= 3
            if exchange != "columns":
                assert eng.comm.calls["all_gather"] >= 9                    # 3 chunks x 3 sweeps
        # delta_stream (opt-in): the scalar all-reduce on a stream of its own, sweeps launched one ahead -- same bits
        runs = []
        for own in (False, True):
            eng = SweepEngine(csr, X, dev, comm=TorchComm(pg, force_collectives=True), exchange="columns", delta_stream=own)
            assert (eng._delta_stream is not None) == own
            eng.build_P()
            ticket, deltas = eng.sweep_launch(gamma), []
            for _ in range(11):
                ahead = eng.sweep_launch(gamma)
                deltas.append(eng.sweep_wait(ticket))
                ticket = ahead
            deltas.append(eng.sweep_wait(ticket))
            runs.append((deltas, eng.get_Z()))
            assert eng.comm.calls["all_reduce"] >= 12
        assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1])
    finally:
        dist.destroy_process_group()


# ---- f4: the README's downstream protocol fed with GPU output ---------------------------------------------------------
def test_f1_harness_on_gpu_embeddings(tmp_path):
    """README.md:51-73 (logistic regression on Z, train ratios 10 % ... 90 %, 10 runs, micro / macro F1) on the
    karate graph with the reference's own label file (tests/data_root/Y, golden G10): embeddings produced by
    Embedder.iterate() on the GPU give the same table as the oracle's embeddings (same splits, same seeds)."""
    spec = importlib.util.spec_from_file_location("evaluate_f1", ROOT / "tools" / "evaluate_f1.py")
    f1 = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(f1)
    kc, lab, g4 = load_golden("g2_karate_csr.npz"), load_golden("g10_karate_labels_log.npz"), load_golden("g4_karate_d16.npz")
    assert list(kc["vertex_ids"]) == list(lab["Y_ids"])
    root = write_data_root(tmp_path / "karate", kc["vertex_ids"], kc["edge_src"], kc["edge_dst"], g4["X"])
    (root / "Y").write_text("\n".join(f"{i}\t{c}" for i, c in zip(lab["Y_ids"], lab["Y_classes"])) + "\n")
    y = f1.read_labels(root / "Y")
    assert len(y) == 34 and len(set(y.tolist())) == 4
    g = Graph(root)
    emb = Embedder(g, CosineSimilarity(), torch.device("cuda"), gamma=float(g4["gamma"]), tolerence=int(g4["tolerence"]),
                   verbose=False)
    emb.iterate()
    Z_gpu = g.Z.numpy()
    orc = O.OracleEmbedder(g.csr.rowptr, g.csr.colidx, torch.from_numpy(g4["X"]), gamma=float(g4["gamma"]),
                           tolerence=int(g4["tolerence"]))
    Z_or = orc.iterate().numpy()
    assert O.rel_l2(torch.from_numpy(Z_gpu), torch.from_numpy(g4["Z_final"])) < 1e-5
    ratios = [r / 10 for r in range(1, 10)]
    t_gpu, t_or, t_ref = (np.array(f1.f1_table(Z, y, ratios, runs=10)) for Z in (Z_gpu, Z_or, g4["Z_final"]))
    assert t_gpu.shape == (9, 3) and np.isfinite(t_gpu).all() and (t_gpu[:, 1:] >= 0).all() and (t_gpu[:, 1:] <= 1).all()
    np.testing.assert_allclose(t_gpu, t_or, rtol=0, atol=1e-12)
    np.testing.assert_allclose(t_gpu, t_ref, rtol=0, atol=1e-12)          # and the reference's own embeddings


def test_plugin_similarity_in_chunks_at_scale(dev):
    """A `batchwise` plug-in callable on a graph with hub rows: the gathered pair batches come 4 MiB at a time, the
    rows above the engine's threshold are soft-maxed by a workgroup each, and P equals the oracle's literal
    gather-and-call (graph.py:119-123); a batch-global callable above the size limit is refused."""
    V, E, d = 50_000, 1_000_000, 64
    csr = synth.rmat_csr(V, E, seed=21)
    X = synth.gaussian_X(V, d, seed=22)
    g = Graph.from_csr(csr, X)
    calls = []

    def dot_sim(a, b):
        calls.append(a.shape[0])
        return (a * b).sum(1) / 8.0
    dot_sim.batchwise = True
    g.PLUGIN_CHUNK_BYTES = 4 << 20
    P = g.build_P(dot_sim)
    assert len(calls) == -(-E // ((4 << 20) // (2 * d * 4))) and sum(calls) == E and max(calls) * 2 * d * 4 <= 4 << 20
    assert g._engine.long_rows[0] is not None and int(np.diff(csr.rowptr).max()) > g._engine.score_threshold
    ref = O.build_P_values(csr.rowptr, csr.colidx, X, similarity=lambda a, b: (a * b).sum(1) / 8.0)
    assert rel(P.values(), ref) < 2e-6
    g.PLUGIN_SINGLE_CALL_MAX_BYTES = 1 << 20
    with pytest.raises(ValueError, match="batchwise = True"):
        g.build_P(lambda a, b: (a * b).sum(1))


def test_loader_sorts_large_edge_lists_on_the_card():
    """csr_from_edges with >= 2^20 edge lines takes the GPU for its one sort: same coalesced adjacency (sorted,
    duplicates merged, self-loops kept: graph.py:104-110) as the oracle's numpy restatement."""
    from clane_amd.graph import GPU_SORT_MIN_EDGES, csr_from_edges
    rng = np.random.default_rng(3)
    V, n = 300_000, 3_000_000
    assert n >= GPU_SORT_MIN_EDGES
    src, dst = rng.integers(0, V, n), rng.integers(0, V, n)
    src[:1000], dst[:1000] = src[1000:2000], dst[1000:2000]          # duplicates
    dst[2000:2100] = src[2000:2100]                                   # self-loops
    got = csr_from_edges(V, src, dst)
    rowptr, colidx = O.build_csr(V, src, dst)
    assert np.array_equal(got.rowptr, rowptr) and np.array_equal(got.colidx, colidx) and got.num_edges < n


def test_workgroups_are_dealt_to_the_xcds_round_robin(dev, k):
    """What the class-affine kernels rely on for their SPEED (never for their results): within one launch workgroup w
    runs on XCD (w + c) % 8, c the same for the whole launch (it carries over from the dispatches before, so it is
    not always 0) -- for small and large grids, 64 to 1024 threads per workgroup, on the default and a side stream.
    All chunks of one class then share one L2 and the eight classes use eight different ones."""
    if torch.cuda.get_device_properties(dev).multi_processor_count != 256:
        pytest.skip("not the 8-XCD / 256-CU partition the class pass is tuned for (results do not depend on it)")
    side = torch.cuda.Stream(dev)
    rotations = set()
    for n, threads in ((8, 64), (64, 256), (2048, 256), (100_000, 256), (20_001, 1024), (333_333, 64)):
        for stream in (torch.cuda.current_stream(dev), side):
            with torch.cuda.stream(stream):
                got = k.xcc_ids(n, threads, dev)
            stream.synchronize()
            assert int(got.min()) >= 0 and int(got.max()) <= 7
            c = int(got[0])
            rotations.add(c)
            want = (torch.arange(n, device=dev, dtype=torch.int32) + c) % 8
            assert torch.equal(got, want), (n, threads, c, int((got != want).sum()))
    print("start XCDs seen:", sorted(rotations))


def test_class_items_in_pieces_on_the_card(dev, k, monkeypatch):
    """The O(E) part of the class-pass layout runs on the card in pieces of rows (bounded scratch, no tensor beyond
    what torch sorts / indexes comfortably): tiny pieces give the items of one piece, and of the host path."""
    from clane_amd import xcd
    csr = synth.rmat_csr(20_000, 600_000, seed=11, device=str(dev))
    X = synth.gaussian_X(20_000, 64, seed=12)
    eng = SweepEngine(csr, X, dev, class_threshold=32, class_phases=2, phase_threshold=128)
    rows = eng.class_rows[0][0].cpu().numpy().astype(np.int64)
    assert rows.size > 100
    args = (eng.local.rowptr, eng.local.colidx, rows, 64, 8)
    kw = dict(phase_threshold=128, phases=2)
    whole = xcd.class_items(*args, colidx_dev=eng.colidx, **kw)
    host = xcd.class_items(*args, **kw)
    monkeypatch.setattr(xcd, "CLASS_ITEMS_PIECE_EDGES", 1000)
    pieces = xcd.class_items(*args, colidx_dev=eng.colidx, **kw)
    for name in whole:
        assert np.array_equal(whole[name], pieces[name]) and np.array_equal(whole[name], host[name]), name


def test_check_csr_refuses_what_would_fault_the_gpu(dev, k):
    """clane_check_csr: one pass on the device over rowptr / colidx before any kernel gathers through them.  A valid CSR
    passes (also an empty one); a decreasing or out-of-range rowptr entry and a column outside the table are named; and
    SweepEngine runs the check itself, so a bad CSR never reaches a gather kernel."""
    csr = synth.rmat_csr(50_000, 1_000_000, seed=21, device=str(dev))
    rp, ci = torch.from_numpy(csr.rowptr).to(dev), torch.from_numpy(csr.colidx).to(dev)
    V, E = csr.num_vertices, csr.num_edges
    k.check_csr(rp, ci, V, E, V)
    k.check_csr(torch.zeros(1, dtype=torch.int64, device=dev), torch.zeros(1, dtype=torch.int32, device=dev), 0, 0, 5)
    with pytest.raises(ValueError, match="colidx holds entries outside"):
        k.check_csr(rp, ci, V, E, V - 1 if int(ci.max()) == V - 1 else int(ci.max()))
    bad = ci.clone()
    bad[E // 2] = -1
    with pytest.raises(ValueError, match="colidx"):
        k.check_csr(rp, bad, V, E, V)
    for where, value in ((V // 3, E + 7), (V // 2, -2), (V, E + 1)):
        bad = rp.clone()
        bad[where] = value
        with pytest.raises(ValueError, match="rowptr"):
            k.check_csr(bad, ci, V, E, V)
    dec = rp.clone()
    i = int(torch.nonzero(rp[1:] > rp[:-1])[5])
    dec[i + 1] = dec[i] - 1 if int(dec[i]) > 0 else dec[i]
    if int(dec[i + 1]) < int(dec[i]):
        with pytest.raises(ValueError, match="rowptr"):
            k.check_csr(dec, ci, V, E, V)
    with pytest.raises(ValueError, match="int64"):
        k.check_csr(rp.to(torch.int32), ci, V, E, V)
    broken = HostCSR(V, csr.rowptr, csr.colidx.copy())
    broken.colidx[123] = V + 5
    with pytest.raises(ValueError, match="colidx holds entries outside"):     # on the host, before anything is uploaded
        SweepEngine(broken, synth.gaussian_X(V, 16, seed=22), dev)
    with pytest.raises(ValueError, match=r"X must be \[V, d\]"):
        SweepEngine(csr, synth.gaussian_X(V - 1, 16, seed=22), dev)


# ---- BASELINE config 1 at "|E| ~ 10k": the symmetrised Cora-shape graph (cyclic: many sweeps per propagate) ---------
def test_corashape_symmetrised_to_convergence(tmp_path):
    """BASELINE configs[1] / SURVEY 8d name the symmetrised variant of the Cora-shape draw (2 708 vertices, 5 429
    directed edges + their reverses = 10 856 distinct edges, d = 1 433 binary BoW: 6 column tiles of a wave).  Unlike
    the DAG-like directed draw it has cycles, so a propagate runs ~100 sweeps before the stopping rule
    (embedder.py:98-108) fires -- the case that exercises that rule on the GPU at this width.  Embedder.iterate()
    through Graph / the C ABI against the oracle's Embedder, final embeddings within 1e-5 (the bar is 1e-4)."""
    g = load_golden("g8_corashape.npz")
    V, d = int(g["V"]), int(g["d"])
    X = torch.zeros(V, d)
    X[torch.from_numpy(g["X_nz_row"].astype(np.int64)), torch.from_numpy(g["X_nz_col"].astype(np.int64))] = 1.0
    src, dst = np.concatenate([g["src"], g["dst"]]), np.concatenate([g["dst"], g["src"]])
    root = write_data_root(tmp_path / "cora_sym", range(V), src, dst, X.numpy())
    graph = Graph(root)
    assert len(graph.E) == 10_858 and graph.csr.num_edges == 10_856        # two pairs were reciprocal already
    assert np.array_equal(graph.csr.indeg(), np.diff(graph.csr.rowptr))    # symmetric
    emb = Embedder(graph, CosineSimilarity(), torch.device("cuda"), gamma=0.76, tolerence=10, verbose=False)
    emb.iterate()
    orc = O.OracleEmbedder(graph.csr.rowptr, graph.csr.colidx, X, gamma=0.76, tolerence=10, plain_c=True)
    Z_or = orc.iterate()
    assert O.rel_l2(graph.Z, Z_or) < 1e-5
    # the first propagate runs ~100 sweeps down to the fp32 noise floor of the delta, where WHICH sweep stops improving
    # is decided by last-ulp noise (seen: 109 on the GPU, 120 in the oracle): the counts agree loosely, what
    # the round did to Z (its outer delta) agrees tightly
    assert orc.sweep_counts[0] > 60 and emb.sweep_counts[0] > 60
    assert abs(emb.sweep_counts[0] - orc.sweep_counts[0]) <= 0.25 * orc.sweep_counts[0]
    assert emb.outer_deltas[0] == pytest.approx(orc.outer_deltas[0], rel=1e-5)
    # and the fixed point itself: z = x + gamma P z with P rebuilt from z (embedder.py:92, graph.py:118-128)
    Zf = graph.Z
    P_f = O.build_P_values(graph.csr.rowptr, graph.csr.colidx, Zf)
    Z_next, _ = O.sweep(graph.csr.rowptr, graph.csr.colidx, P_f, X, Zf, 0.76)
    assert O.rel_l2(Z_next, Zf) < 1e-5


# ---- a mega-hub row in the default suite: 1M+ edges through the class pass and its many-slot combine ----------------
def test_mega_hub_row_through_class_pass(dev):
    """One row of 1 050 000 edges (and one of 300 000) at d = 256 fp32 over a sparse background of 1.1M vertices:
    the hub takes the XCD-affine pass as ~4 100 chunks whose partial sums spmm_class_combine_kernel adds in slot order,
    and edge_softmax_class_kernel combines as many {max, sum} pairs (embedder.py:90-92 and graph.py:122-123 have no
    degree limit).  build_P and two sweeps against the C oracle, which builds its own P; the hub rows on their own;
    and in per-edge mode (where the softmax moves with every score) the hub's P against float64."""
    from oracle import clane_oracle_c as OC
    V, d, gamma = 1_100_000, 256, 0.76
    rng = np.random.default_rng(77)
    deg = rng.integers(0, 4, size=V)                      # background: 0..3 edges, a quarter of the rows are sinks
    hubs = {12_345: 1_050_000, 777_777: 300_000}
    for r, n in hubs.items():
        deg[r] = n
    rowptr = np.zeros(V + 1, dtype=np.int64)
    np.cumsum(deg, out=rowptr[1:])
    cols = rng.integers(0, V, size=int(rowptr[-1])).astype(np.int32)
    for r, n in hubs.items():
        cols[rowptr[r]:rowptr[r + 1]] = rng.choice(V, size=n, replace=False)
    # rows sorted by column, duplicates inside a short row made distinct (a CSR row holds a column once)
    order = np.lexsort((cols, np.repeat(np.arange(V), deg)))
    cols = cols[order]
    dup = np.nonzero((cols[1:] == cols[:-1]) & (np.repeat(np.arange(V), deg)[1:] == np.repeat(np.arange(V), deg)[:-1]))[0]
    keep = np.ones(cols.size, dtype=bool)
    keep[dup + 1] = False
    row_of = np.repeat(np.arange(V), deg)[keep]
    cols = cols[keep]
    rowptr = np.zeros(V + 1, dtype=np.int64)
    np.cumsum(np.bincount(row_of, minlength=V), out=rowptr[1:])
    csr = HostCSR(V, rowptr, cols)
    E = csr.num_edges
    X = synth.gaussian_X(V, d, seed=78)
    eng = SweepEngine(csr, X, dev)
    # destinations are spread evenly here (the 32 768 rows the eight L2s hold take ~6 % of the reads), so only rows that
    # need their work spread take the class pass (engine.hot_read_share): the two hubs
    assert not eng.class_affinity and eng.hot_read_share < 0.2 and eng.class_threshold == 4096
    assert eng.class_rows[0] is not None and eng.class_rows[0][0].numel() == 2
    assert eng.class_slots[0] >= (1_050_000 + 300_000) // eng.class_chunk       # thousands of slots behind two rows
    eng.build_P()
    P = eng.P_global()
    P_or, _ = OC.build_P(csr.rowptr, csr.colidx, X)
    assert P.numel() == E and O.rel_l2(P, P_or) < 2e-6
    for r in hubs:
        a, b = rowptr[r], rowptr[r + 1]
        assert O.rel_l2(P[a:b], P_or[a:b]) < 2e-6 and float(P[a:b].double().sum()) == pytest.approx(1.0, abs=1e-5)
    Z_or = X
    for _ in range(2):
        delta = eng.sweep(gamma)
        Z_or, delta_or = OC.sweep(csr.rowptr, csr.colidx, P_or, X, Z_or, gamma)
        assert delta == pytest.approx(delta_or, rel=1e-6)
    Z = eng.get_Z()
    assert O.rel_l2(Z, Z_or) < 1e-6
    for r in hubs:
        assert O.rel_l2(Z[r], Z_or[r]) < 2e-6
    sink = torch.from_numpy(np.diff(rowptr) == 0)
    assert torch.equal(Z[sink], X[sink])
    # bitwise repeatable: the slot order alone fixes the association of the 4 000 partial sums
    eng_b = SweepEngine(csr, X, dev)
    eng_b.build_P()
    assert torch.equal(eng_b.P, eng.P)
    for _ in range(2):
        eng_b.sweep(gamma)
    assert torch.equal(eng_b.Zcur, eng.Zcur)
    del eng_b, eng
    # per-edge cosine: the hub's scores differ edge by edge, its softmax is no longer 1/deg
    eng_p = SweepEngine(csr, X, dev, cosine_mode="per_edge")
    eng_p.build_P()
    P_pe = eng_p.P_global()
    want = per_edge_rows_f64(csr, X, list(hubs) + [5, 6, 7, 8])
    for r, w in want.items():
        a, b = rowptr[r], rowptr[r + 1]
        if b > a:
            assert np.linalg.norm(P_pe[a:b].numpy() - w) <= 2e-6 * np.linalg.norm(w), r
    hub = P_pe[rowptr[12_345]:rowptr[12_346]]
    assert float(hub.max() / hub.min()) > 1.2                    # the softmax did move


# ---- the driver's N = 8 shape, rehearsed with as many ranks as one card may hold ------------------------------------
def test_bench_four_ranks_share_one_gpu_with_an_idle_column_rank():
    """`python bench.py --gpus N` as the driver starts it (no launcher), with more ranks than a row has 16-byte packs:
    `tiny12` (d = 12 fp32 = 3 packs) over 4 ranks leaves rank 3 without columns -- the shape of karate-sized inputs on
    an 8-GPU node.  4 ranks, not 8: a GPU box admits 6 processes on its card, and this test process and the launcher
    count (5 ranks were killed by the box's process guard).  The record must describe the group the collectives ran in
    (`comm`), carry north_star's literal division beside the default one (`north_star_literal`), and both must match
    the C oracle."""
    import json
    import os
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    run = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "4", "--backend", "gloo", "--share-gpu",
                          "--workload", "tiny12", "--exchange", "columns", "--steps", "3", "--warmup", "1", "--blocks",
                          "3", "--also-exchange", "allgather_all,allgather,halo"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 4 and r["blocks"] == 3 and "column split x4" in r["config"]["parallelism"]
    probe = r["comm"]["fabric_probe"]               # four child processes measured the literal plan's message first
    assert probe["microbench"]["world"] == 4 and probe["microbench"]["all_gather_correct"] and probe["rccl_log"] is None
    assert "all_gather_vs_direct_exchange" in probe and "matrices" in probe["topology"]
    assert r["time_plan"]["worst_case_s"] < r["time_plan"]["driver_limit_s"] and "total_s" in r["time_plan"]["spent_s"]
    assert r["parity_rel_l2_vs_oracle_after_1_sweep"] < 1e-5 and r["parity_P_rel_l2_vs_oracle"] < 2e-6
    comm = r["comm"]
    assert comm["ranks_seen"] == 4 and len(comm["devices"]) == 4 and len(comm["ms_per_step_by_rank"]) == 4
    assert comm["backend"] == "gloo" and comm["exchange"] == "columns" and comm["exchange_bytes_per_sweep"] == 0
    assert all(dv["name"] == comm["devices"][0]["name"] for dv in comm["devices"])
    lit = r["north_star_literal"]
    assert "error" not in lit and lit["exchange"] == "allgather_all" and lit["comm"]["ranks_seen"] == 4
    assert lit["parity_rel_l2_vs_oracle_after_1_sweep"] < 1e-5 and lit["comm"]["exchange_bytes_per_sweep"] > 0
    assert len(lit["comm"]["exchange_bytes_per_sweep_by_rank"]) == 4
    assert lit["last_delta"] == pytest.approx(r["last_delta"], rel=5e-5)
    # ... and the leaner row splits, so that one run on real links compares every division
    others = r["other_divisions"]
    assert set(others) == {"allgather", "halo"}
    for name, blk in others.items():
        assert "error" not in blk and blk["comm"]["exchange"] == name and blk["comm"]["ranks_seen"] == 4
        assert blk["parity_rel_l2_vs_oracle_after_1_sweep"] < 1e-5 and blk["value"] > 0
        assert blk["comm"]["exchange_bytes_per_sweep"] <= lit["comm"]["exchange_bytes_per_sweep"]
        assert blk["last_delta"] == pytest.approx(r["last_delta"], rel=5e-5)


# ---- the row-binning heuristics on graphs off their tuning set (results must not depend on them) --------------------
@pytest.mark.parametrize("family", ["regular", "uniform", "stars"])
@pytest.mark.parametrize("dtype,d", [(torch.float32, 256), (torch.bfloat16, 128)])
def test_thresholds_off_the_tuning_set(dev, family, dtype, d):
    """The thresholds that deal rows to kernels were swept on R-MAT and one power-law draw (engine.py, xcd.py).  Here:
    a near-regular graph (every row 48..80 edges: at 1-KiB rows ALL rows sit around the class threshold of 64), a
    uniform random one, and a star-heavy one (rows that read every vertex).  With the default thresholds, with the
    class pass off and with the class threshold moved, build_P and two sweeps match the C oracle -- speed is
    tools/threshold_robustness.py's business (profiles/r03_threshold_robustness.md), results are this test's."""
    from oracle import clane_oracle_c as OC
    V = 60_000
    csr = {"regular": lambda: synth.regular_csr(V, 48, 80, device=str(dev)),
           "uniform": lambda: synth.uniform_random_csr(V, 1_200_000, device=str(dev)),
           "stars": lambda: synth.star_csr(V, 4, V, device=str(dev))}[family]()
    X = synth.gaussian_X(V, d, seed=31).to(dtype)
    Xf = X.float()
    P_or, _ = OC.build_P(csr.rowptr, csr.colidx, Xf)
    Z_or = Xf
    for _ in range(2):
        Z_or, _ = OC.sweep(csr.rowptr, csr.colidx, P_or, Xf, Z_or, 0.76)
    tol_p, tol_z = (1e-4, 8e-3) if dtype == torch.bfloat16 else (2e-6, 1e-6)
    seen = set()
    for kw in ({}, {"class_threshold": 0}, {"class_threshold": 32}, {"class_threshold": 512}, {"long_threshold": 16}):
        eng = SweepEngine(csr, X, dev, **kw)
        seen.add((eng.class_threshold, eng.long_threshold, eng.class_rows[0] is not None))
        eng.build_P()
        assert O.rel_l2(eng.P_global().float(), P_or) < tol_p, (family, kw)
        for _ in range(2):
            eng.sweep(0.76)
        assert O.rel_l2(eng.get_Z().float(), Z_or) < tol_z, (family, kw)
    assert len(seen) >= 4                       # the variants really took different kernels


def test_bench_main_record_survives_a_stuck_literal_block():
    """The `north_star_literal` block runs after the main measurement and must never cost it: past its deadline every
    rank prints / leaves on its own (a rank stuck in a collective cannot be talked to).  A deadline of a millisecond
    stands in for a hung all-gather: ONE JSON line still comes out, with the main division's numbers, its parity and
    `comm` block, and an `error` in place of the literal block -- and the launcher sees a NON-ZERO exit: a division that
    hung is a failure the driver must be able to key on, the record says which."""
    import json
    import os
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["CLANE_BENCH_LITERAL_DEADLINE_S"] = "0.001"
    run = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu",
                          "--workload", "tiny", "--steps", "3", "--warmup", "1", "--blocks", "2", "--no-fabric-probe"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert run.returncode != 0, run.stderr[-3000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["value"] > 0 and r["parity_rel_l2_vs_oracle_after_1_sweep"] < 1e-5
    assert r["comm"]["ranks_seen"] == 2 and "no result within" in r["also_exchange_error"]
    assert "value" not in r["north_star_literal"]


def test_bench_n_gpu_flow_through_real_rccl_with_one_rank():
    """`bench.py --rehearse-rccl`: the whole N > 1 flow of bench.py -- `init_process_group("nccl")`, the input agreement
    check (`all_gather_object`), the column split's collectives (all-reduce of the partial dot products, of the norms,
    of the delta), barriers and MAX-reductions of the timing protocol, the `comm` block, and the `north_star_literal`
    block with its in-place `all_gather_into_tensor` per chunk issued async from side streams -- through the REAL
    RCCL library with the one rank this box can give it.  Each collective is the identity, so the numbers must be the
    plain one-GPU run's; what this catches are API / dtype / stream mistakes that gloo accepts."""
    import json
    import os
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    common = ["--workload", "tiny", "--steps", "3", "--warmup", "1", "--blocks", "2", "--no-cpu-baseline"]
    plain = subprocess.run([sys.executable, str(ROOT / "bench.py")] + common, capture_output=True, text=True, timeout=600,
                           cwd=ROOT, env=env)
    assert plain.returncode == 0, plain.stderr[-2000:]
    run = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--rehearse-rccl"] + common, capture_output=True,
                         text=True, timeout=600, cwd=ROOT, env=env)
    assert run.returncode == 0, run.stderr[-3000:]
    r0 = json.loads([ln for ln in plain.stdout.splitlines() if ln.startswith("{")][-1])
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert "rehearsal" in r and r["n_gpus"] == 1 and "cpu_baseline" not in r
    comm = r["comm"]
    assert comm["backend"] == "nccl" and comm["ranks_seen"] == 1 and comm["exchange"] == "columns"
    probe = comm["fabric_probe"]                    # RCCL's own log of the probe's one-rank group, parsed
    assert probe["child_rc"] == 0 and probe["rccl_log"]["init"] == {"nranks": 1} and probe["rccl_log"]["channels"]["coll"] > 0
    assert "RCCL" in probe["rccl_log"]["version"] and probe["microbench"]["all_gather_correct"]
    assert comm["collectives_issued"]["all_reduce"] > 10 and comm["collective_detail"]["sweeps_timed"] > 0
    assert r["parity_rel_l2_vs_oracle_after_1_sweep"] < 1e-5 and r["parity_P_rel_l2_vs_oracle"] < 2e-6
    assert r["last_delta"] == pytest.approx(r0["last_delta"], rel=5e-5)
    lit = r["north_star_literal"]
    assert "error" not in lit and lit["exchange"] == "allgather_all" and lit["comm"]["backend"] == "nccl"
    assert lit["comm"]["collectives_issued"]["all_gather"] >= 3 * 2 + 2             # one in-place all-gather per sweep
    assert lit["parity_rel_l2_vs_oracle_after_1_sweep"] < 1e-5
    assert lit["last_delta"] == pytest.approx(r0["last_delta"], rel=5e-5)
    assert set(r["other_divisions"]) == {"allgather"} and "error" not in r["other_divisions"]["allgather"]


# ---- float64 content embeddings (a `C.npy` of dtype float64, reference golden G6) at config 2's size ----------------
def test_config2_shape_in_float64(dev):
    """The reference computes in whatever dtype `C.npy` holds (graph.py:52-58: a float64 file gives float64 embeddings,
    golden G6).  Config 2's graph (R-MAT 200k / 4M) with d = 128 float64 -- 1-KiB rows: the one-wave-per-row kernels,
    the XCD-affine class pass and its combine, K1's class softmax, all in double -- `build_P` and three sweeps through
    the engine against the PyTorch-CPU oracle in float64, to 1e-13; per-edge mode on the hub rows against the float64
    restatement; bitwise repeatability."""
    V, E, d, gamma = 200_000, 4_000_000, 128, 0.76
    csr = synth.rmat_csr(V, E, seed=1, device=str(dev))
    X = synth.gaussian_X(V, d, seed=2).double() + 1e-9 * synth.gaussian_X(V, d, seed=3).double()   # not fp32-representable
    eng = SweepEngine(csr, X, dev)
    assert eng.dtype == torch.float64 and eng.acc_dtype == torch.float64 and eng.class_rows[0] is not None
    eng.build_P()
    P = eng.P_global()
    P_or = O.build_P_values(csr.rowptr, csr.colidx, X)
    assert P.dtype == torch.float64 and O.rel_l2(P, P_or) < 1e-13
    Ps = O.as_sparse(csr.rowptr, csr.colidx, P_or)
    Z_or = X
    for _ in range(3):
        delta = eng.sweep(gamma)
        Z_or, delta_or = O.sweep(csr.rowptr, csr.colidx, P_or, X, Z_or, gamma, Ps)
        assert delta == pytest.approx(float(delta_or), rel=1e-12)
    Z = eng.get_Z()
    assert Z.dtype == torch.float64 and O.rel_l2(Z, Z_or) < 1e-13
    eng_b = SweepEngine(csr, X, dev)
    eng_b.build_P()
    for _ in range(3):
        eng_b.sweep(gamma)
    assert torch.equal(eng_b.P, eng.P) and torch.equal(eng_b.Zcur, eng.Zcur)
    del eng_b
    eng_p = SweepEngine(csr, X, dev, cosine_mode="per_edge")
    eng_p.build_P()
    P_pe = eng_p.P_global().numpy()
    hubs = np.argsort(np.diff(csr.rowptr))[-3:]
    for r, want in per_edge_rows_f64(csr, X, list(hubs) + [11, 12, 13]).items():
        a, b = csr.rowptr[r], csr.rowptr[r + 1]
        if b > a:
            assert np.linalg.norm(P_pe[a:b] - want) <= 1e-12 * np.linalg.norm(want), r
