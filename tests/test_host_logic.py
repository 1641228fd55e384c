"""Host-side logic of clane_amd on CPU: file loader, CSR, object-model facade, row partition,
SweepEngine launch sequence and Embedder control flow.  The kernels are replaced by the
oracle-backed test double (tests/oracle_kernels.py) -- only here, only by injection."""
import io
import contextlib
from pathlib import Path

import numpy as np
import pytest
import torch

from clane_amd import _hip
from clane_amd.embedder import Embedder, IterativeEmbedder
from clane_amd.engine import SweepEngine
from clane_amd.graph import Graph
from clane_amd.partition import HostCSR, RowPartition, localize
from clane_amd.xcd import xcd_class
from clane_amd.similarity import AsymmertricSimilarity, CosineSimilarity, Similarity
from oracle import clane_oracle as O

from .conftest import load_golden, write_data_root
from .oracle_kernels import OracleKernels

KARATE_LIKE = ["g4_karate_d2.npz", "g4_karate_d16.npz", "g5_symkarate_d16_g0.5.npz",
               "g5_symkarate_d16_g0.76.npz", "g5_symkarate_d2_g0.76.npz", "g7_readme5.npz", "g6_tiny_f64.npz",
               "g11_hubs320_d8_g0.9.npz", "g12_hubs150_d256_g0.76.npz"]


def graph_from_golden(tmp_path, name):
    g = load_golden(name)
    if "edge_src" in g.files:
        vids = g["vertex_ids"] if "vertex_ids" in g.files else load_golden("g2_karate_csr.npz")["vertex_ids"]
        src, dst = g["edge_src"], g["edge_dst"]
    else:
        k = load_golden("g2_karate_csr.npz")
        vids, src, dst = k["vertex_ids"], k["edge_src"], k["edge_dst"]
    root = write_data_root(tmp_path / name, vids, src, dst, g["X"])
    return g, Graph(root)


def attach_cpu_engine(graph, **kw):
    eng = SweepEngine(graph.csr, graph.X, "cpu", OracleKernels(), **kw)
    graph._attach_engine(eng)
    return eng


# ---- reference tests/test_graph.py, re-expressed ----------------------------------------------
def test_load_zachary(karate_root):
    g = Graph(data_root=karate_root, embedding_dim=16)
    assert len(g.vertex_ids) == 34 and len(g.V) == 34 and len(g.E) == 78 and len(g) == 34
    for v in g.V:
        assert isinstance(v.x, torch.Tensor) and v.x.shape[-1] == 16
    assert g.d == 16 and g.dispense_pair is False and g[5] == 5


def test_build_A_and_get_nbrs(karate_root):
    g = Graph(data_root=karate_root, embedding_dim=16)
    A = g.A
    assert A.shape == (34, 34) and A.is_coalesced()
    np.testing.assert_array_equal(A.indices().numpy(), load_golden("g2_karate_csr.npz")["A_indices"])
    assert g.get_nbrs(33).tolist() == [8, 9, 13, 14, 15, 18, 19, 20, 22, 23, 26, 27, 28, 29, 30, 31, 32]
    for v in g.V:
        nb = g.get_nbrs(v.idx)
        assert nb.dim() == 1 and nb.dtype == torch.int64
    # object-model facade: per-line adjacency lists, edge endpoints
    e0 = g.E[0]
    assert (e0.src.id_, e0.dst.id_) == ("2", "1") and e0.src.idx == 1
    assert g.V[33].outgoing_indices == g.get_nbrs(33).tolist()
    assert 33 in g.V[8].incoming_indices


def test_duplicates_selfloop_string_ids_f64(tmp_path):
    gold, g = graph_from_golden(tmp_path, "g6_tiny_f64.npz")
    assert len(g.E) == int(gold["num_E"]) == 5 and g.csr.num_edges == 4
    assert g.X.dtype == torch.float64 and g.d == 128 and g.X.shape == (4, 3)   # embedding_dim kept, shape from C.npy
    assert g.get_nbrs(0).tolist() == [1, 2] and g.get_nbrs(2).tolist() == [] and g.get_nbrs(3).tolist() == [3]
    assert g.V[0].outgoing_indices == [1, 2, 1]          # per-line list keeps the duplicate (graph.py:82)


def test_graph_dtype_option(karate_root):
    assert Graph(karate_root, embedding_dim=4).X.dtype == torch.float32            # default: as upstream
    assert Graph(karate_root, embedding_dim=4, dtype="bfloat16").X.dtype == torch.bfloat16
    assert Graph(karate_root, embedding_dim=4, dtype=torch.float64).X.dtype == torch.float64
    with pytest.raises(ValueError):
        Graph(karate_root, embedding_dim=4, dtype="int8")


def test_loader_errors(tmp_path):
    with pytest.raises(FileNotFoundError):
        Graph(tmp_path / "nope")
    root = write_data_root(tmp_path / "noE", ["1", "2"], ["1"], ["2"])
    (root / "E").unlink()
    with pytest.raises(FileNotFoundError):
        Graph(root)
    with pytest.raises(ValueError):
        Graph(write_data_root(tmp_path / "unk", ["1", "2"], ["1"], ["3"]))
    bad = write_data_root(tmp_path / "bad", ["1", "2"], ["1"], ["2"])
    (bad / "E").write_text("1 2\n")
    with pytest.raises(ValueError):
        Graph(bad)


def test_first_occurrence_wins_for_duplicate_ids(tmp_path):
    g = Graph(write_data_root(tmp_path / "dup", ["a", "b", "a"], ["b"], ["a"]), embedding_dim=2)
    assert g.get_nbrs(1).tolist() == [0]


def test_Z_is_fresh_and_set_Z_roundtrip(karate_root):
    g = Graph(karate_root, embedding_dim=4)
    z = g.Z
    assert torch.equal(z, g.X) and z.data_ptr() != g.X.data_ptr()
    z.zero_()
    assert torch.equal(g.Z, g.X)
    new = torch.arange(34 * 4, dtype=torch.float32).reshape(34, 4)
    g.set_Z(new)
    assert torch.equal(g.Z, new) and torch.equal(g.V[3].z, new[3])
    g.V[3].z = torch.ones(4)
    assert torch.equal(g.Z[3], torch.ones(4)) and torch.equal(g.Z[4], new[4])
    eng = attach_cpu_engine(g)
    assert torch.equal(eng.get_Z(), g.Z) and torch.equal(g.Z[3], torch.ones(4))


# ---- no GPU => loud failure, never a fallback ---------------------------------------------------
@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_product_path_refuses_to_run_without_gpu(karate_root):
    g = Graph(karate_root, embedding_dim=4)
    with pytest.raises(_hip.ClaneHipError, match="no CPU fallback"):
        g.build_P(CosineSimilarity())
    with pytest.raises(_hip.ClaneHipError):
        Embedder(g, CosineSimilarity(), torch.device("cpu")).iterate()
    with pytest.raises(_hip.ClaneHipError):
        CosineSimilarity()(torch.ones(3), torch.ones(3))
    with pytest.raises(_hip.ClaneHipError, match="GPU memory"):
        _hip.kernels().row_sqnorm(torch.ones(2, 4), 4, torch.zeros(2))


# ---- plugin surface -------------------------------------------------------------------------
def test_similarity_plugin_surface():
    import clane_amd.similarity as S
    assert getattr(S, "CosineSimilarity")(foo="bar").is_trainable() is False
    assert isinstance(CosineSimilarity(), Similarity) and CosineSimilarity(mode="per_edge").mode == "per_edge"
    with pytest.raises(ValueError):
        CosineSimilarity(mode="nope")
    a = AsymmertricSimilarity(n_dim=4)
    assert a.is_trainable() and hasattr(a, "parameters")
    x, y = torch.rand(3, 4), torch.rand(3, 4)
    assert a(x, y).shape == (3,) and not torch.allclose(a(x, y), a(y, x))
    with pytest.raises(NotImplementedError):
        IterativeEmbedder()


# ---- partition ------------------------------------------------------------------------------
@pytest.mark.parametrize("V,W,C", [(34, 1, 1), (34, 1, 3), (34, 2, 1), (34, 2, 3), (5, 4, 2), (1000, 8, 4),
                                   (34, 8, 1), (34, 8, 2), (34, 8, 4), (320, 8, 4), (7, 8, 2)])
@pytest.mark.parametrize("with_mask", [False, True])
def test_partition_is_a_bijection_with_in_place_allgather_spans(V, W, C, with_mask):
    live = (np.random.default_rng(V + W + C).random(V) < 0.4) if with_mask else None
    parts = [RowPartition.create(V, W, r, C, live_mask=live, seed=3) for r in range(W)]
    p0 = parts[0]
    pos = p0.position_of_vertex()
    assert len(np.unique(pos)) == V and pos.max() < p0.padded_vertices and pos.min() >= 0
    owned = np.concatenate([p.local_positions() for p in parts])
    assert sorted(owned.tolist()) == list(range(p0.padded_vertices))          # every position has exactly one owner
    if with_mask and W > 1:                                                     # live rows sit in the exchanged region
        assert (pos[live] < p0.live_total).all() and (pos[~live] >= p0.live_total).all()
    covered = np.zeros(p0.padded_vertices, dtype=int)
    for r, p in enumerate(parts):
        lp = p.local_positions()
        blocks = p.blocks()
        assert sum(b.nrows for b in blocks) == p.n_local
        for b in blocks:
            np.testing.assert_array_equal(lp[b.local_start:b.local_start + b.nrows], np.arange(b.row0, b.row0 + b.nrows))
            if W == 1:
                assert b.span is None
            elif b.span is not None:                                             # in-place all-gather form
                assert (b.span[1] - b.span[0]) == W * b.nrows and b.row0 == b.span[0] + r * b.nrows
                if r == 0:
                    covered[b.span[0]:b.span[1]] += 1
    if W > 1:
        assert (covered[:p0.live_total] == 1).all() and (covered[p0.live_total:] == 0).all()
        qs = p0.quiet_span()
        assert qs is None or (qs[0] == p0.live_total and qs[1] == p0.padded_vertices and qs[1] - qs[0] == W * qs[2])


def test_localize_relabels_and_keeps_every_edge():
    gold = load_golden("g5_symkarate_d16_g0.76.npz")
    idx = gold["A_indices"]
    rowptr, colidx = O.build_csr(34, idx[0], idx[1])
    csr = HostCSR(34, rowptr, colidx)
    seen = np.zeros(csr.num_edges, dtype=int)
    for r in range(3):
        part = RowPartition.create(34, 3, r, 2, live_mask=csr.live_mask(), seed=1)
        loc = localize(csr, part)
        pos = part.position_of_vertex()
        seen[loc.edge_origin] += 1
        for l in range(part.n_local):
            v = loc.vertex[l]
            cols = loc.colidx[loc.rowptr[l]:loc.rowptr[l + 1]]
            if v < 0:
                assert len(cols) == 0
                continue
            assert sorted(cols.tolist()) == cols.tolist()
            assert sorted(cols.tolist()) == sorted(pos[colidx[rowptr[v]:rowptr[v + 1]]].tolist())
            assert loc.indeg[l] == np.sum(colidx == v)
    assert (seen == 1).all()


# ---- engine launch sequence vs goldens ----------------------------------------------------------
@pytest.mark.parametrize("name", KARATE_LIKE)
@pytest.mark.parametrize("chunks,shuffle", [(1, False), (3, True)])
def test_engine_build_P_and_first_sweep(tmp_path, name, chunks, shuffle):
    gold, g = graph_from_golden(tmp_path, name)
    attach_cpu_engine(g, chunks=chunks, shuffle=shuffle, seed=5)
    P = g.build_P(CosineSimilarity())
    assert P.is_coalesced() and P.shape == (len(g), len(g)) and P.dtype == g.X.dtype
    np.testing.assert_array_equal(P.indices().numpy(), gold["A_indices"])
    np.testing.assert_allclose(P.values().numpy(), gold["P0_values"], rtol=3e-6, atol=1e-7)
    dense = P.to_dense().sum(1)
    sink = np.diff(g.csr.rowptr) == 0
    assert torch.allclose(dense[~torch.from_numpy(sink)], torch.ones((~sink).sum(), dtype=dense.dtype), atol=1e-5)
    assert float(dense[torch.from_numpy(sink)].abs().sum()) == 0.0
    eng = g._engine
    delta = eng.sweep(float(gold["gamma"]))
    Z1 = torch.from_numpy(gold["Z_sweep1"])
    assert O.rel_l2(g.Z, Z1) < 1e-6
    assert delta == pytest.approx(float((Z1 - torch.from_numpy(gold["X"])).abs().sum()), rel=1e-5)


@pytest.mark.parametrize("name", KARATE_LIKE)
def test_embedder_iterate_matches_reference_final_Z(tmp_path, name):
    gold, g = graph_from_golden(tmp_path, name)
    attach_cpu_engine(g)
    emb = Embedder(g, CosineSimilarity(), torch.device("cpu"), gamma=float(gold["gamma"]),
                   tolerence=int(gold["tolerence"]), save_history=True, verbose=False)
    emb.iterate()
    assert O.rel_l2(g.Z, torch.from_numpy(gold["Z_final"])) < 1e-6
    assert emb.tolerences["global"].value == 0 and emb.tolerences["propagation"].value == 0
    assert len(emb.history["Z"]) == len(emb.sweep_counts) == len(emb.outer_deltas)
    assert [len(h) for h in emb.history["Z"]] == emb.sweep_counts
    # sweep counts depend on last-ulp noise near the fixed point: bounded, never pinned
    assert abs(emb.sweep_counts[0] - gold["sweep_counts"][0]) <= max(3, 0.15 * gold["sweep_counts"][0])
    assert O.rel_l2(emb.history["Z"][0][0], torch.from_numpy(gold["Z_sweep1"])) < 1e-6
    assert O.rel_l2(emb.history["Z"][0][-1], torch.from_numpy(gold["Z_prop1"])) < 1e-6
    assert emb.minimum_amount_updated_Z == min(emb.outer_deltas)


def test_column_slices_and_auto_exchange():
    from clane_amd.engine import column_slice, pick_exchange
    for d, dtype, W in [(256, torch.float32, 8), (128, torch.bfloat16, 8), (100, torch.float32, 3), (2, torch.float32, 4),
                        (1433, torch.float32, 8), (64, torch.float64, 5)]:
        vec = {torch.float32: 4, torch.float64: 2, torch.bfloat16: 8}[dtype]
        cuts = [column_slice(d, dtype, W, r) for r in range(W)]
        assert cuts[0][0] == 0 and cuts[-1][1] == d and all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
        assert all(c0 % vec == 0 for c0, c1 in cuts if c1 > c0)                       # whole 16-byte packs
        widths = [c1 - c0 for c0, c1 in cuts]
        assert sum(widths) == d and max(widths) - min(widths) < 2 * vec                  # even up to one pack
    # config 3: 1 KiB rows -> 128-byte slices at 8 GPUs: columns; config 4: 256-byte bf16 rows -> halo from 8 GPUs on
    assert pick_exchange(256, torch.float32, 8) == "columns" and pick_exchange(256, torch.float32, 16) == "columns"
    assert pick_exchange(128, torch.bfloat16, 4) == "columns" and pick_exchange(128, torch.bfloat16, 8) == "halo"
    assert pick_exchange(2, torch.float32, 2) == "halo"
    with pytest.raises(ValueError, match="exchange must be"):           # the 2-D division of round 4 is gone
        SweepEngine(HostCSR(2, np.array([0, 1, 1]), np.array([1], dtype=np.int32)), torch.zeros(2, 4), "cpu",
                    OracleKernels(), exchange="grid")
    with pytest.raises(ValueError, match="exchange must be"):
        SweepEngine(HostCSR(2, np.array([0, 1, 1]), np.array([1], dtype=np.int32)), torch.zeros(2, 4), "cpu",
                    OracleKernels(), exchange="rows")


def test_idle_sweeps_are_accounted_but_not_launched(tmp_path):
    """At the fp32 fixed point a sweep's delta is exactly 0 and stays 0 while P is frozen: Embedder runs the
    reference's countdown over those sweeps without launching them.  Decisions, counts, printout, history and
    result are the same as with every sweep launched."""
    runs = {}
    for skip in (True, False):
        root = write_data_root(tmp_path / f"one{skip}", ["a"], ["a"], ["a"], np.array([[1.0, -2.0, 0.5]], dtype=np.float32))
        g = Graph(root)
        attach_cpu_engine(g)
        emb = Embedder(g, CosineSimilarity(), torch.device("cpu"), gamma=0.5, tolerence=4, save_history=True,
                       skip_idle_sweeps=skip)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            emb.iterate()
        runs[skip] = (emb, g.Z, buf.getvalue())
    (a, Za, out_a), (b, Zb, out_b) = runs[True], runs[False]
    assert a.sweep_counts == b.sweep_counts and a.outer_deltas == b.outer_deltas and out_a == out_b
    assert torch.equal(Za, Zb) and torch.equal(Za, torch.tensor([[2.0, -4.0, 1.0]]))
    # round 1 ends with `tolerence` idle sweeps; round 2 launches one sweep (delta 0) and is idle; rounds 3.. replay it
    assert b.sweeps_launched == sum(b.sweep_counts) and a.sweeps_launched == a.sweep_counts[0] - 4 + 1
    assert a.outer_deltas[-1] == 0.0 and a.sweep_counts[-1] == 4 + 1      # one launched sweep + the countdown
    for ha, hb in zip(a.history["Z"], b.history["Z"]):
        assert len(ha) == len(hb) and all(torch.equal(x, y) for x, y in zip(ha, hb))


@pytest.mark.parametrize("world", [2, 3])
def test_halo_p2p_places_rows_in_the_readers_tables(tmp_path, world):
    """exchange="halo_p2p" on CPU: ranks as threads, each engine's kernels (the oracle-backed double) store finished
    rows into the OTHER engines' tables through the places the layouts of all ranks imply; no exchange call is
    made during sweeps, and every rank reproduces the single-process oracle."""
    import threading
    from .thread_comm import ThreadWorld
    gold, g = graph_from_golden(tmp_path, "g5_symkarate_d16_g0.76.npz")
    X = torch.from_numpy(gold["X"])
    P_or = O.build_P_values(g.csr.rowptr, g.csr.colidx, X)
    Z_or, deltas_or = X.clone(), []
    for _ in range(4):
        Z_or, dl = O.sweep(g.csr.rowptr, g.csr.colidx, P_or, X, Z_or, 0.76)
        deltas_or.append(float(dl))
    shared, results, errors = ThreadWorld(world), [None] * world, []

    def run(rank):
        try:
            comm = shared.comm(rank)
            calls = []
            real = comm.all_to_all_rows
            comm.all_to_all_rows = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
            eng = SweepEngine(g.csr, X, "cpu", OracleKernels(), comm=comm, chunks=2, exchange="halo_p2p", seed=5)
            assert eng.p2p and eng.halo and all(b is None or b.shape[0] == 0 for b in eng.send_buf)
            eng.build_P()
            deltas = [eng.sweep(0.76) for _ in range(4)]
            results[rank] = (eng.get_Z(), deltas, len(calls))
        except Exception as exc:
            errors.append((rank, exc))
            shared.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors
    for Z, deltas, n_exchanges in results:
        assert n_exchanges == 0 and O.rel_l2(Z, Z_or) < 1e-6
        assert deltas == pytest.approx(deltas_or, rel=1e-5)


def test_every_division_on_random_graphs():
    """24 seeded random graphs (with empty rows, hubs, vertices nobody reads) x world 2..5 x 1..4 chunks x every way
    of dividing the work -- ranks as threads, oracle-backed kernels: build_P, three sweeps and the read-out agree
    with the single-process oracle on every rank."""
    import threading
    from .thread_comm import ThreadWorld
    rng = np.random.default_rng(77)
    modes = ["columns", "halo", "halo_p2p", "allgather", "allgather_all"]
    for case in range(24):
        V = int(rng.integers(6, 60))
        world, chunks, mode = int(rng.integers(2, 6)), int(rng.integers(1, 5)), modes[case % len(modes)]
        d = int(rng.choice([3, 8, 20]))
        deg = rng.integers(0, min(V, 9), size=V)
        deg[rng.random(V) < 0.25] = 0
        deg[int(rng.integers(V))] = V                                    # one hub that points at everybody
        cols = [np.sort(rng.choice(V, size=k, replace=False)) for k in deg]
        rowptr = np.zeros(V + 1, dtype=np.int64)
        np.cumsum(deg, out=rowptr[1:])
        csr = HostCSR(V, rowptr, np.concatenate(cols + [np.empty(0, int)]).astype(np.int32))
        X = torch.from_numpy(rng.standard_normal((V, d)).astype(np.float32))
        P_or = O.build_P_values(csr.rowptr, csr.colidx, X)
        Z_or, deltas_or = X.clone(), []
        for _ in range(3):
            Z_or, dl = O.sweep(csr.rowptr, csr.colidx, P_or, X, Z_or, 0.7)
            deltas_or.append(float(dl))
        shared, results, errors = ThreadWorld(world), [None] * world, []

        def run(rank):
            try:
                # every other case: rows above 3 edges take the XCD-affine class pass (any division)
                ct = 3 if case % 2 else None
                eng = SweepEngine(csr, X, "cpu", OracleKernels(), comm=shared.comm(rank), chunks=chunks, exchange=mode,
                                  seed=case, class_threshold=ct, class_chunk=64)
                eng.build_P()
                P_mine = torch.zeros(csr.num_edges)
                P_mine[torch.from_numpy(eng.local.edge_origin)] = eng.P[:eng.E_loc]
                owned = torch.zeros(csr.num_edges, dtype=torch.bool)
                owned[torch.from_numpy(eng.local.edge_origin)] = True
                deltas = [eng.sweep(0.7) for _ in range(3)]
                results[rank] = (eng.get_Z(), deltas, P_mine, owned)
            except Exception as exc:
                errors.append((rank, exc))
                shared.barrier.abort()

        threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=120)
        tag = f"case {case}: V={V} world={world} chunks={chunks} {mode} d={d}"
        assert not errors, (tag, errors)
        covered = torch.zeros(csr.num_edges, dtype=torch.int32)
        for Z, deltas, P_mine, owned in results:
            assert O.rel_l2(Z, Z_or) < 2e-6, tag
            assert deltas == pytest.approx(deltas_or, rel=1e-5), tag
            assert torch.allclose(P_mine[owned], P_or[owned].float(), rtol=1e-5, atol=1e-7), tag
            covered += owned.int()
        assert bool((covered >= 1).all()), tag                           # every edge's P is held by some rank


@pytest.mark.parametrize("name", ["g5_symkarate_d16_g0.5.npz", "g4_karate_d2.npz", "g7_readme5.npz"])
def test_lagged_check_keeps_the_stopping_rule(tmp_path, name):
    """Embedder(lagged_check=True): the next sweep is launched before this sweep's delta is read, and dropped when
    the rule says stop.  Same counts, deltas, printout and embeddings as the synchronous loop, bit for bit."""
    runs = []
    for lagged in (False, True):
        gold, g = graph_from_golden(tmp_path / str(lagged), name)
        eng = attach_cpu_engine(g)
        emb = Embedder(g, CosineSimilarity(), torch.device("cpu"), gamma=float(gold["gamma"]), tolerence=4,
                       lagged_check=lagged)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            emb.iterate()
        runs.append((emb.sweep_counts, emb.outer_deltas, buf.getvalue(), g.Z, emb.sweeps_launched, eng.sweeps_done))
    a, b = runs
    assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2] and torch.equal(a[3], b[3])
    assert a[4] == b[4] == a[5] == b[5]                # discarded launches are not counted as sweeps


def test_history_sink_receives_every_sweep_in_order(tmp_path):
    """Embedder(history_sink=...): the same embeddings as history["Z"], streamed (outer, sweep, Z) in order from
    the writer thread and not retained; a failing sink surfaces at flush."""
    gold, g = graph_from_golden(tmp_path, "g5_symkarate_d16_g0.5.npz")
    attach_cpu_engine(g)
    kept = Embedder(g, CosineSimilarity(), torch.device("cpu"), gamma=0.5, tolerence=3, save_history=True,
                    verbose=False)
    kept.iterate()
    gold, g2 = graph_from_golden(tmp_path / "b", "g5_symkarate_d16_g0.5.npz")
    attach_cpu_engine(g2)
    got = []
    emb = Embedder(g2, CosineSimilarity(), torch.device("cpu"), gamma=0.5, tolerence=3, save_history=True,
                   verbose=False, history_sink=lambda o, s, Z: got.append((o, s, Z)))
    emb.iterate()
    assert emb.sweep_counts == kept.sweep_counts and emb.history["Z"] == [[] for _ in emb.sweep_counts]
    assert [(o, s) for o, s, _ in got] == [(o, s) for o, n in enumerate(kept.sweep_counts) for s in range(n)]
    for o, s, Z in got:
        assert torch.equal(Z, kept.history["Z"][o][s])

    def broken(o, s, Z):
        raise OSError("disk full")
    gold, g3 = graph_from_golden(tmp_path / "c", "g7_readme5.npz")
    attach_cpu_engine(g3)
    emb = Embedder(g3, CosineSimilarity(), torch.device("cpu"), tolerence=2, save_history=True, verbose=False,
                   history_sink=broken)
    with pytest.raises(RuntimeError, match="history sink failed"):
        emb.iterate()


def test_history_parts_of_several_ranks_are_put_together(tmp_path):
    """--save_history on several GPUs: every rank's writer thread saves the part of Z it staged, rank 0's writer puts
    the parts of a sweep together for the sink, in order, and leaves no part behind.  Three "ranks" as writer threads
    of one process: two column slices and... a row split with padding rows; parts arrive late and out of order."""
    import threading
    import time
    from clane_amd.embedder import _HistoryWriter, _PartsAssembler
    from clane_amd.engine import StagedZ, place_piece
    V, d, W = 11, 6, 3
    gen = torch.Generator().manual_seed(0)
    sweeps = [torch.randn(V, d, generator=gen) for _ in range(5)]

    def pieces_of(Z, kind):
        if kind == "columns":
            cuts = [(0, 2), (2, 4), (4, 6)]
            return [{"kind": "columns", "c0": a, "c1": b} | {"Z": Z[:, a:b].clone()} for a, b in cuts]
        owner = [torch.tensor([0, 3, 6, 9]), torch.tensor([1, 4, 7, 10]), torch.tensor([2, 5, 8, -1])]   # -1: padding
        return [{"kind": "rows", "vertex": v, "Z": torch.where(v[:, None] >= 0, Z[v.clamp_min(0)], torch.full((1, d), 9.0))}
                for v in owner]

    for kind in ("columns", "rows"):
        got = []
        writers = []
        for r in range(W):
            w = _HistoryWriter((lambda o, s, Z: got.append((o, s, Z))) if r == 0 else (lambda o, s, Z: 1 / 0))
            w.assembler = _PartsAssembler(tmp_path / kind, r, W, (V, d), torch.float32)
            writers.append(w)

        def feed(r, delay):
            for i, Z in enumerate(sweeps):
                time.sleep(delay)
                piece = pieces_of(Z, kind)[r]
                st = StagedZ(ready=piece.pop("Z"), where=piece)
                writers[r].submit(i // 3, i % 3, st)
        threads = [threading.Thread(target=feed, args=(r, 0.01 * (W - r))) for r in range(W)]   # rank 0 is the slowest
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for w in writers:
            w.flush()                       # the other ranks' sinks (1 / 0) were never called: only rank 0 gets matrices
        assert [(o, s) for o, s, _ in got] == [(i // 3, i % 3) for i in range(5)]
        for (_, _, Z), want in zip(got, sweeps):
            assert torch.equal(Z, want)
        assert list((tmp_path / kind).iterdir()) == []
    full = torch.zeros(V, d)
    for piece in pieces_of(sweeps[0], "rows"):
        place_piece(full, piece)
    assert torch.equal(full, sweeps[0])
    with pytest.raises(RuntimeError, match="piece"):
        StagedZ(ready=sweeps[0], where={"kind": "columns", "c0": 0, "c1": d}).result()
    with pytest.raises(RuntimeError, match="result"):
        StagedZ(ready=sweeps[0]).piece()


def test_class_pass_follows_the_read_skew():
    """XCD affinity can only turn SKEWED gathers into L2 hits: `hot_read_share` = the share of all edge reads going to
    the rows the eight 4-MiB L2s hold between them.  A skewed graph keeps the tuned class threshold; one whose
    destinations are spread evenly only sends rows above HEAVY_ROW_EDGES through the pass (as the hub splitter), and
    its 33..64-edge rows do not inherit the class threshold as their one-wave limit; an explicit threshold is obeyed."""
    from clane_amd import engine as E
    V, d = 30_000, 2048                         # 8-KiB rows: the L2s hold 4 096 of them
    rng = np.random.default_rng(3)
    deg = rng.integers(40, 81, size=V)
    rowptr = np.zeros(V + 1, dtype=np.int64)
    np.cumsum(deg, out=rowptr[1:])
    even = HostCSR(V, rowptr, rng.integers(0, V, size=int(rowptr[-1])).astype(np.int32))
    share = E.hot_read_share(even, d * 4)
    assert 4096 / V <= share < E.MIN_HOT_READ_SHARE
    hot = rng.integers(0, 500, size=int(rowptr[-1]))            # every edge reads one of 500 rows
    skewed = HostCSR(V, rowptr, hot.astype(np.int32))
    assert E.hot_read_share(skewed, d * 4) == 1.0
    assert E.hot_read_share(even, 64) == 1.0                    # the whole table fits
    X = torch.zeros(V, d)
    kern = OracleKernels()
    e_even = SweepEngine(even, X, "cpu", kernels=kern)
    assert not e_even.class_affinity and e_even.class_threshold == E.HEAVY_ROW_EDGES and e_even.class_rows[0] is None
    assert e_even.long_threshold == E.UNSKEWED_LONG_THRESHOLD                   # 128: no hubs, one wave per row (r04)
    e_skew = SweepEngine(skewed, X, "cpu", kernels=kern)
    assert e_skew.class_affinity and e_skew.class_threshold == 64 and e_skew.class_rows[0] is not None
    assert e_skew.long_threshold == 64
    e_forced = SweepEngine(even, X, "cpu", kernels=kern, class_threshold=64)
    assert e_forced.class_threshold == 64 and e_forced.class_rows[0] is not None
    assert e_even.kernel_config()["class_affinity"] is False and e_skew.kernel_config()["class_affinity"] is True


def test_kernel_backend_contract_is_enforced():
    """Every call the engine makes on its kernel object is an abstract method of ``KernelBackend``: an implementation
    that lacks one cannot be instantiated (so nothing -- the CSR check, say -- is skipped silently), the engine does
    not probe its kernel object for optional methods, and the binding and the test double implement the same list."""
    import inspect
    from clane_amd._hip import HipKernels, KernelBackend
    from .oracle_kernels import OracleKernels
    need = KernelBackend.__abstractmethods__
    assert {"check_csr", "spmm_update", "edge_score", "edge_score_class", "make_mirror", "build_info"} <= need
    assert not HipKernels.__abstractmethods__ and not OracleKernels.__abstractmethods__

    class Lacking(OracleKernels):
        check_csr = KernelBackend.check_csr         # abstract again
    Lacking.__abstractmethods__ = frozenset({"check_csr"})
    with pytest.raises(TypeError, match="abstract"):
        Lacking()
    import clane_amd.engine as E
    src = inspect.getsource(E)
    assert "hasattr(self.k" not in src and "getattr(self.k" not in src
    # what the engine calls on self.k is all in the contract
    import re
    called = set(re.findall(r"(?:self\.k|\bk)\.([a-z_0-9]+)\(", src)) | set(re.findall(r'_bind\("([a-z_]+)"', src))
    assert called and called <= need | {"bind"}, called - need


def test_embedder_prints_delta_and_tolerance_per_sweep(tmp_path):
    gold, g = graph_from_golden(tmp_path, "g7_readme5.npz")
    attach_cpu_engine(g)
    emb = Embedder(g, CosineSimilarity(), torch.device("cpu"), tolerence=2)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        emb.propagate()
    lines = buf.getvalue().strip().splitlines()
    assert len(lines) == emb.sweep_counts[0] and lines[-1].endswith(" 0") and lines[0].endswith(" 2")


def test_per_edge_mode_is_true_cosine(tmp_path):
    gold, g = graph_from_golden(tmp_path, "g5_symkarate_d16_g0.76.npz")
    attach_cpu_engine(g)
    P = g.build_P(CosineSimilarity(mode="per_edge"))
    rowptr, colidx = g.csr.rowptr, g.csr.colidx
    ref = O.build_P_values(rowptr, colidx, g.X, mode="per_edge")
    np.testing.assert_allclose(P.values().numpy(), ref.numpy(), rtol=1e-5)
    assert not np.allclose(P.values().numpy(), gold["P0_values"], rtol=1e-3)


def test_custom_similarity_callable_goes_through_plugin_protocol(tmp_path):
    gold, g = graph_from_golden(tmp_path, "g5_symkarate_d2_g0.76.npz")
    attach_cpu_engine(g)
    calls = []

    def sim(a, b):
        calls.append((tuple(a.shape), tuple(b.shape)))
        return O.cosine_similarity(a, b)

    P = g.build_P(sim)
    assert calls == [((156, 2), (156, 2))]                 # ONE batched call with all edges (graph.py:121)
    np.testing.assert_allclose(P.values().numpy(), gold["P0_values"], rtol=3e-6, atol=1e-7)

    # a callable that declares itself `batchwise` (a pair's score ignores the rest of the batch) is fed in chunks
    def dot(a, b):
        calls.append((tuple(a.shape), tuple(b.shape)))
        return (a * b).sum(1)
    whole = g.build_P(dot).values()
    dot.batchwise = True
    del calls[:]
    g.PLUGIN_CHUNK_BYTES = 2 * 2 * 4 * 50                  # 50 edge pairs of d=2 fp32 per chunk
    np.testing.assert_array_equal(g.build_P(dot).values().numpy(), whole.numpy())
    assert [c[0][0] for c in calls] == [50, 50, 50, 6]
    # a batch-global callable larger than the limit is refused with the way out in the message
    g.PLUGIN_SINGLE_CALL_MAX_BYTES = 1000
    with pytest.raises(ValueError, match="batchwise = True"):
        g.build_P(sim)


# ---- native loader (csrc/host_loader.cpp) == Python loader == reference semantics --------------------
def test_native_loader_matches_python_loader(tmp_path, karate_root):
    from clane_amd import graph as G
    assert G._host_lib() is not None, "libclane_host.so not built (run __graft_entry__.build())"
    cases = {"karate": karate_root,
             "strings": write_data_root(tmp_path / "s", ["a", "b b", "a", "", " d"], ["a", "b b", "", " d"],
                                        ["b b", "a", "a", "b b"])}
    crlf = write_data_root(tmp_path / "crlf", ["x", "y", "z"], ["x", "z"], ["y", "x"])
    for f in ("V", "E"):
        (crlf / f).write_bytes((crlf / f).read_bytes().replace(b"\n", b"\r\n"))
    cases["crlf"] = crlf
    rng = np.random.default_rng(0)
    ids = [f"v{i}" for i in rng.permutation(5000)]
    cases["big"] = write_data_root(tmp_path / "big", ids, rng.choice(ids, 40000), rng.choice(ids, 40000))
    for name, root in cases.items():
        vids = G.read_vertex_ids(root)
        ns, nd = G._native_parse_edges(Path(root))
        ps, pd_ = G._python_parse_edges(Path(root), vids)
        np.testing.assert_array_equal(ns, ps, err_msg=name)
        np.testing.assert_array_equal(nd, pd_, err_msg=name)
    for bad_lines, exc in (("1 2\n", ValueError), ("1\t2\t3\n", ValueError), ("1\t9\n", ValueError), ("", ValueError)):
        root = write_data_root(tmp_path / f"bad{abs(hash(bad_lines))}", ["1", "2"], ["1"], ["2"])
        (root / "E").write_text(bad_lines)
        with pytest.raises(exc):
            G._native_parse_edges(root)
        with pytest.raises(exc):
            G._python_parse_edges(root, ["1", "2"])


def test_native_loader_multithreaded_pieces(tmp_path):
    """An E file above 4 MiB is cut at line boundaries and parsed by several threads: same indices as line-by-line
    parsing, and of several bad lines the FIRST one in the file is the one reported."""
    from clane_amd import graph as G
    if G._host_lib() is None:
        pytest.skip("libclane_host.so not built")
    rng = np.random.default_rng(3)
    n_v, n_e = 5000, 150_000
    ids = [f"vertex-with-a-long-name-{i:07d}" for i in range(n_v)]
    src, dst = rng.integers(0, n_v, n_e), rng.integers(0, n_v, n_e)
    lines = [f"{ids[a]}\t{ids[b]}" for a, b in zip(src, dst)]
    root = tmp_path / "big"
    root.mkdir()
    (root / "V").write_text("\n".join(ids) + "\n")
    (root / "E").write_text("\n".join(lines) + "\n\n")
    assert (root / "E").stat().st_size > (1 << 22)
    s2, d2 = G._native_parse_edges(root)
    assert np.array_equal(s2, src) and np.array_equal(d2, dst)
    bad = list(lines)
    bad[140_000] = "no-tab-here"
    bad[90_000] = f"{ids[1]}\tnobody"
    bad[120_000] = "a\tb\tc"
    (root / "E").write_text("\n".join(bad))
    with pytest.raises(ValueError, match="'nobody' is not in list"):
        G._native_parse_edges(root)
    bad[20_000] = "a\tb\tc"
    (root / "E").write_text("\n".join(bad))
    with pytest.raises(ValueError, match="E line 20001"):
        G._native_parse_edges(root)


def test_clane_import_shim():
    import clane.graph, clane.similarity, clane.embedder                    # noqa: E401
    from clane.__main__ import get_parser
    import clane_amd.graph
    assert clane.graph.Graph is clane_amd.graph.Graph
    assert clane.similarity.CosineSimilarity().is_trainable() is False
    assert clane.embedder.Embedder.Tolerence(3).value == 3 and get_parser().prog == "clane"


# ---- halo layout: sender and receiver agree on every (source, destination, chunk) list ----------------
@pytest.mark.parametrize("W,C", [(2, 1), (3, 2), (8, 4)])
def test_halo_layout_is_consistent_across_ranks(W, C):
    from clane_amd.halo import build_halo_layout
    from clane_amd import synth
    csr = synth.rmat_csr(3000, 30000, seed=5, device="cpu")
    outdeg = csr.outdeg()
    L = [build_halo_layout(csr, W, r, C, seed=7) for r in range(W)]
    n_local = L[0].n_local
    assert all(l.n_local == n_local and np.array_equal(l.vertex_slot, L[0].vertex_slot) for l in L)
    owned = np.concatenate([l.table_vertex[:n_local][l.table_vertex[:n_local] >= 0] for l in L])
    assert sorted(owned.tolist()) == list(range(3000))                        # every vertex owned exactly once
    for r, l in enumerate(L):
        own = l.table_vertex[:n_local]
        # every column this rank reads is in its table, and CSR rows match the global graph
        for lr in range(n_local):
            v = own[lr]
            cols = l.table_vertex[l.local.colidx[l.local.rowptr[lr]:l.local.rowptr[lr + 1]]]
            ref = csr.colidx[csr.rowptr[v]:csr.rowptr[v + 1]] if v >= 0 else []
            assert sorted(cols.tolist()) == sorted(np.asarray(ref).tolist())
        assert len(l.blocks) == C and sum(b.nrows for b in l.blocks) == n_local
        for c, b in enumerate(l.blocks):
            ex = b.exchange
            assert len(ex.in_splits) == len(ex.out_splits) == W and ex.in_splits[r] == 0 and ex.out_splits[r] == 0
            assert sum(ex.in_splits) == ex.send_rows.size and sum(ex.out_splits) == ex.recv_rows
            assert ((ex.send_rows >= b.local_start) & (ex.send_rows < b.local_start + b.nrows)).all()
            assert (outdeg[own[ex.send_rows]] > 0).all()                     # constant rows are never re-sent
            # what rank r sends to q in chunk c is exactly what q expects from r in its chunk-c halo slice
            off = 0
            for q in range(W):
                sent = own[ex.send_rows[off:off + ex.in_splits[q]]]
                off += ex.in_splits[q]
                exq = L[q].blocks[c].exchange
                start = exq.recv_start + sum(exq.out_splits[:r])
                expect = L[q].table_vertex[start:start + exq.out_splits[r]]
                np.testing.assert_array_equal(sent, expect)
        # the constant tail of the halo holds remote rows without out-edges
        tail = l.table_vertex[l.blocks[-1].exchange.recv_start + l.blocks[-1].exchange.recv_rows:]
        assert (outdeg[tail] == 0).all()


def test_edge_cache_roundtrip_and_invalidation(tmp_path, karate_root):
    g1 = Graph(karate_root, embedding_dim=4, cache=True)
    assert (karate_root / ".clane_edges.npz").exists()
    g2 = Graph(karate_root, embedding_dim=4, cache=True)                    # served from the cache
    np.testing.assert_array_equal(g1.csr.colidx, g2.csr.colidx)
    assert len(g2.E) == 78
    e = (karate_root / "E").read_text()
    (karate_root / "E").write_text(e + "1\t34\n")                           # file changed -> cache ignored
    g3 = Graph(karate_root, embedding_dim=4, cache=True)
    assert len(g3.E) == 79 and g3.get_nbrs(0).tolist() == [33]


def test_f1_harness_runs(tmp_path):
    import importlib.util
    spec = importlib.util.spec_from_file_location("evaluate_f1", Path(__file__).parent.parent / "tools" / "evaluate_f1.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.default_rng(0)
    y = rng.integers(0, 3, 300)
    Z = np.eye(3)[y] + 0.1 * rng.standard_normal((300, 3))                   # separable embeddings -> F1 ~ 1
    rows = mod.f1_table(Z, y, [0.1, 0.5, 0.9], runs=2)
    assert len(rows) == 3 and all(m > 0.95 and M > 0.95 for _, m, M in rows)
    (tmp_path / "Y").write_text("\n".join(f"{i}\t{'ABC'[c]}" for i, c in enumerate(y)) + "\n")
    assert np.array_equal(mod.read_labels(tmp_path / "Y"), y)


def test_per_sweep_log_lines_are_the_references(tmp_path):
    """Golden G10: the reference prints the 0-d delta tensor and the countdown after every sweep
    (embedder.py:104), e.g. ``tensor(25.7074) 10``; the same lines come out here."""
    import re
    gold, g = graph_from_golden(tmp_path, "g4_karate_d2.npz")
    attach_cpu_engine(g)
    want = list(load_golden("g10_karate_labels_log.npz")["propagate_stdout"])
    emb = Embedder(g, CosineSimilarity(), torch.device("cpu"), gamma=float(gold["gamma"]),
                   tolerence=int(gold["tolerence"]))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        emb.propagate()
    got = buf.getvalue().splitlines()
    assert got[:6] == want[:6]                       # deltas well above fp32 noise: identical text
    assert all(re.fullmatch(r"tensor\([0-9.e+-]+\) \d+", ln) for ln in got)
    assert got[-1].endswith(" 0") and abs(len(got) - len(want)) <= 3


def test_mega_rows_items_are_scheduled_by_column():
    """Rows one of whose (phase, class) segments alone is more than an L2 holds (`mega_segment_edges`): their work items
    come first in their class and in the order of their first column, so that chunks of different mega rows gathering
    the same stretch of the table run next to each other; the other rows' items keep row order behind them; slots (the
    order of every sum) and the items themselves are what they were."""
    from clane_amd.xcd import class_items, xcd_class
    rng = np.random.default_rng(9)
    V = 4000
    deg = np.full(6, 40)
    deg[[1, 4]] = V                                   # two rows that read every vertex
    cols = []
    for k in deg:
        c = np.sort(rng.choice(V, size=k, replace=False))
        cols.append(c[np.lexsort((c, xcd_class(c)))])            # (class, column) order, as the engine lays class rows out
    rowptr = np.zeros(7, dtype=np.int64)
    np.cumsum(deg, out=rowptr[1:])
    colidx = np.concatenate(cols).astype(np.int32)
    rows = np.arange(6)
    plain = class_items(rowptr, colidx, rows, 64, 8)
    mega = class_items(rowptr, colidx, rows, 64, 8, mega_segment_edges=100,    # 4000 / 8 = 500 edges a class > 100
                       mega_min_edges=V // 4)
    sparse_hubs = class_items(rowptr, colidx, rows, 64, 8, mega_segment_edges=100, mega_min_edges=2 * V)
    assert all(np.array_equal(plain[k], sparse_hubs[k]) for k in plain)         # long rows that share little: row order
    assert np.array_equal(plain["slot_ptr"], mega["slot_ptr"])
    key = lambda it: sorted(zip(it["e0"][it["len"] > 0].tolist(), it["len"][it["len"] > 0].tolist(),   # noqa: E731
                                it["slot"][it["len"] > 0].tolist(), it["row"][it["len"] > 0].tolist()))
    assert key(plain) == key(mega)                    # the same items with the same slots, in another order
    ipb = mega["items_per_block"]
    for c in range(8):
        blocks = [b for b in range(len(mega["e0"]) // ipb) if b % 8 == c]
        seq = [i for b in blocks for i in range(b * ipb, (b + 1) * ipb) if mega["len"][i] > 0]
        is_mega = np.isin(mega["row"][seq], [1, 4])
        n_mega = int(is_mega.sum())
        assert n_mega > 0 and is_mega[:n_mega].all() and not is_mega[n_mega:].any()       # mega rows first
        first_col = colidx[mega["e0"][seq[:n_mega]]]
        assert (np.diff(first_col) >= 0).all() and len(set(mega["row"][seq[:n_mega]])) == 2   # by column, rows interleaved
        rest_rows = mega["row"][seq[n_mega:]]
        assert (np.diff(rest_rows) >= 0).all()                                          # the others: row by row
        assert bool((xcd_class(colidx[mega["e0"][seq]].astype(np.int64)) == c).all())


@pytest.mark.parametrize("chunks,hot", [(1, True), (3, True), (1, False)])
def test_class_affine_rows_layout_and_result(chunks, hot):
    """Rows above `class_threshold` edges: edges sorted by (XCD class of the column, column), cut into chunks of one class, chunk
    blocks of class b at block index 8 j + b (the test double asserts that contract), slots contiguous per row --
    and the sweep equals the oracle's whatever the thresholds."""
    from clane_amd.xcd import class_items
    rng = np.random.default_rng(5)
    V, d = 700, 12
    deg = rng.integers(0, 9, size=V)
    deg[rng.random(V) < 0.2] = 0
    for i, h in enumerate((700, 300, 129, 65, 64, 9)):
        deg[11 * i + 5] = h
    cols = [np.sort(rng.choice(V, size=k, replace=False)) for k in deg]
    rowptr = np.zeros(V + 1, dtype=np.int64)
    np.cumsum(deg, out=rowptr[1:])
    csr = HostCSR(V, rowptr, np.concatenate(cols).astype(np.int32))
    X = torch.from_numpy(rng.standard_normal((V, d)).astype(np.float32))
    P_or = O.build_P_values(csr.rowptr, csr.colidx, X)
    Z_or, deltas_or = X.clone(), []
    for _ in range(3):
        Z_or, dl = O.sweep(csr.rowptr, csr.colidx, P_or, X, Z_or, 0.7)
        deltas_or.append(float(dl))
    for ct, chunk, lt, phases in ((8, 64, None, 1), (64, 128, 4, 2), (128, 64, 0, 4), (8, 64, None, 4)):
        eng = SweepEngine(csr, X, "cpu", OracleKernels(), chunks=chunks, hot_rows_first=hot, class_threshold=ct,
                          class_chunk=chunk, long_threshold=lt, class_phases=phases, phase_threshold=200)
        n_class = sum(0 if c is None else c[0].numel() for c in eng.class_rows)
        assert n_class == int((deg > ct).sum()) and 0 < eng.long_threshold <= ct
        assert eng.kernel_names()["split"].startswith("spmm_class") and eng.kernel_config()["class_threshold"] == ct
        # layout: the class rows' edges are sorted by (class, column); everybody else's by column
        lr, lc = eng.local.rowptr, eng.local.colidx.astype(np.int64)
        for r in range(eng.part.n_local):
            c = lc[lr[r]:lr[r + 1]]
            heavy = phases > 1 and c.size > max(200, ct)
            sub = xcd_class(c) + (8 * ((c >> 6) % phases) if heavy else 0)
            key = sub * 10**9 + c if c.size > ct else c
            assert (np.diff(key) > 0).all()
        if phases > 1:          # the heavy rows' blocks come first, phase by phase; the test double checks the classes
            assert eng.kernel_config()["class_phases"] == phases and eng.phase_threshold == max(200, ct)
        eng.build_P()
        assert O.rel_l2(eng.P_global(), P_or) < 1e-6
        deltas = [eng.sweep(0.7) for _ in range(3)]
        assert O.rel_l2(eng.get_Z(), Z_or) < 2e-6 and deltas == pytest.approx(deltas_or, rel=1e-5)
        kb = eng.kernel_bytes()
        assert kb["split"] > 0 and sum(kb.values()) == sum(SweepEngine(csr, X, "cpu", OracleKernels(), chunks=chunks,
                                                                       class_threshold=0).kernel_bytes().values())
    with pytest.raises(ValueError, match="class_chunk"):
        SweepEngine(csr, X, "cpu", OracleKernels(), class_threshold=8, class_chunk=100)
    # the edge order has a two-pass form for key ranges beyond int64: same permutation
    from clane_amd.partition import edge_order
    g = torch.Generator().manual_seed(0)
    rows_t = torch.sort(torch.randint(0, 50, (4000,), generator=g)).values
    cols_t = torch.randperm(4000, generator=g) % 977 + 977 * (torch.arange(4000) % 4)        # unique per row: distinct values
    cls_t = ((cols_t >> 3) & 7) * (rows_t % 2)
    for c in (None, cls_t):
        one, two = edge_order(rows_t, cols_t, c, 4000, two_pass=False), edge_order(rows_t, cols_t, c, 4000, two_pass=True)
        k = rows_t * 10**8 + (0 if c is None else c) * 10**5 + cols_t
        assert torch.equal(k[one], k[two]) and bool((k[one][1:] >= k[one][:-1]).all())
    # the item builder refuses rows that are not in class order
    with pytest.raises(AssertionError, match="sorted by"):
        class_items(csr.rowptr, csr.colidx, np.array([5]), 64, 8)


def test_preprocessing_in_pieces_gives_the_same_layout(monkeypatch):
    """Graphs beyond one torch sort (2^31 - 1 elements) are laid out in pieces of rows / halves of the key range:
    with tiny piece sizes the result is the one-piece result."""
    from clane_amd import graph as G, partition, synth, xcd
    from clane_amd.partition import RowPartition, localize
    csr = synth.rmat_csr(3000, 60000, seed=1, device="cpu")
    for world, rank, chunks in ((1, 0, 1), (3, 1, 2)):
        part = RowPartition.create(3000, world, rank, chunks, live_mask=csr.live_mask(), priority=csr.indeg(),
                                   shuffle=False)
        for kw in (dict(), dict(class_threshold=16, phase_threshold=64, phases=2)):
            whole = localize(csr, part, None, **kw)
            monkeypatch.setattr(partition, "LOCALIZE_PIECE_EDGES", 37)
            pieces = localize(csr, part, None, **kw)
            monkeypatch.undo()
            for name in ("rowptr", "colidx", "edge_origin", "vertex", "indeg"):
                assert np.array_equal(getattr(whole, name), getattr(pieces, name)), (world, kw, name)
    assert list(xcd.row_pieces(np.array([0, 3, 3, 10, 11, 11, 30, 31]), 8)) == [(0, 2), (2, 5), (5, 6), (6, 7)]
    assert list(xcd.row_pieces(np.array([0, 0, 0]), 8)) == [(0, 2)]
    rng = np.random.default_rng(0)
    src, dst = rng.integers(0, 50, 5000), rng.integers(0, 50, 5000)
    whole = G.csr_from_edges(50, src, dst)
    monkeypatch.setattr(G, "SORT_MAX_ELEMENTS", 100)
    halves = G.csr_from_edges(50, src, dst)
    assert np.array_equal(whole.rowptr, halves.rowptr) and np.array_equal(whole.colidx, halves.colidx)


def test_host_native_code_is_clean_under_asan_and_ubsan():
    """tests/sanitize_host.sh: the V/E parser (16 threads) and the C oracle (OpenMP) under AddressSanitizer + UBSan,
    the parser's threads under ThreadSanitizer."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    root = Path(__file__).resolve().parent.parent
    run = subprocess.run([str(root / "tests" / "sanitize_host.sh")], capture_output=True, text=True, timeout=300)
    if "cannot find -lasan" in run.stderr or "cannot find -lubsan" in run.stderr:
        pytest.skip("sanitizer runtimes not installed")
    if "cannot find -ltsan" in run.stderr:
        pytest.skip("sanitizer runtimes not installed")
    assert run.returncode == 0 and run.stdout.count("sanitize_host: ok") == 2, run.stdout[-2000:] + run.stderr[-4000:]


def test_native_loader_strips_what_python_strips(tmp_path):
    """`read().strip()` (graph.py:44, :73) removes every str.isspace() character -- 0x1c-0x1f and the Unicode spaces
    too, and a lone '\\r' reads as '\\n' -- while the same characters INSIDE the file stay part of the ids: the native
    parser and the line-by-line Python restatement agree on 300 seeded files."""
    import random
    from clane_amd import graph as G
    if G._host_lib() is None:
        pytest.skip("libclane_host.so not built")
    ws = [chr(c) for c in (9, 10, 11, 12, 13, 0x1c, 0x1d, 0x1e, 0x1f, 32, 0x85, 0xa0, 0x1680, 0x2000, 0x2003, 0x200a,
                           0x2028, 0x2029, 0x202f, 0x205f, 0x3000)]
    assert all(c.isspace() for c in ws)
    rng = random.Random(0)
    ids = ["a", "bé", "c c", "中", "e e", "　x"]
    for trial in range(300):
        pad = lambda n: "".join(rng.choice(ws) for _ in range(rng.randint(0, n)))      # noqa: E731
        (tmp_path / "V").write_text(pad(3) + "\n".join(ids) + pad(3), encoding="utf-8", newline="")
        (tmp_path / "E").write_text(pad(2) + "\n".join(f"{rng.choice(ids)}\t{rng.choice(ids)}" for _ in range(6)) + pad(2),
                                    encoding="utf-8", newline="")
        outcome = []
        for parse in (lambda: G._python_parse_edges(tmp_path, G.read_vertex_ids(tmp_path)),
                      lambda: G._native_parse_edges(tmp_path)):
            try:
                s, d = parse()
                outcome.append((s.tolist(), d.tolist()))
            except Exception as e:      # noqa: BLE001 -- both must fail alike (e.g. an id lost to the strip)
                outcome.append(type(e).__name__)
        assert outcome[0] == outcome[1], (trial, outcome)


def test_files_that_are_not_utf8_raise_what_text_mode_raises(tmp_path):
    """Upstream opens V and E in text mode (graph.py:43, :72): bytes that are not UTF-8 are a UnicodeDecodeError before
    any line is looked at.  The native parser recognises them and leaves the raising to the Python loop."""
    from clane_amd import graph as G
    (tmp_path / "V").write_bytes(b"a\nb\nc")
    (tmp_path / "E").write_bytes(b"zz\tb\nb\t\xff\n")            # line 1 has an unknown id, line 2 a bad byte
    with pytest.raises(UnicodeDecodeError):
        G.read_edge_indices(tmp_path, G.read_vertex_ids(tmp_path))
    (tmp_path / "E").write_bytes(b"a\tb\nb\t\xed\xa0\x80\n")      # a UTF-16 surrogate: overlong / surrogate forms are errors too
    with pytest.raises(UnicodeDecodeError):
        G.read_edge_indices(tmp_path, G.read_vertex_ids(tmp_path))
    (tmp_path / "V").write_text("é\n中\n😀\nz", encoding="utf-8")
    (tmp_path / "E").write_text("é\t😀\n中\tz", encoding="utf-8")
    src, dst = G.read_edge_indices(tmp_path, G.read_vertex_ids(tmp_path))
    assert src.tolist() == [0, 1] and dst.tolist() == [2, 3]


def test_host_csr_validate():
    """What SweepEngine checks on the host before anything is indexed on the device."""
    from clane_amd.partition import HostCSR
    good = HostCSR(3, np.array([0, 2, 2, 3], dtype=np.int64), np.array([1, 2, 0], dtype=np.int32))
    good.validate()
    HostCSR(1, np.array([0, 0], dtype=np.int64), np.empty(0, dtype=np.int32)).validate()
    for rowptr, colidx, what in (([0, 2, 1, 3], [1, 2, 0], "rowptr"), ([1, 2, 2, 3], [1, 2, 0], "rowptr"),
                                 ([0, 2, 2, 4], [1, 2, 0], "rowptr"), ([0, 2, 2, 3], [1, 3, 0], "colidx"),
                                 ([0, 2, 2, 3], [1, -1, 0], "colidx")):
        with pytest.raises(ValueError, match=what):
            HostCSR(3, np.array(rowptr, dtype=np.int64), np.array(colidx, dtype=np.int32)).validate()
    with pytest.raises(ValueError, match="int64"):
        HostCSR(3, np.array([0, 2, 2, 3], dtype=np.int32), np.array([1, 2, 0], dtype=np.int32)).validate()
    with pytest.raises(ValueError, match="colidx holds entries outside"):
        SweepEngine(HostCSR(3, np.array([0, 2, 2, 3], dtype=np.int64), np.array([1, 7, 0], dtype=np.int32)),
                    torch.zeros(3, 4), "cpu", kernels=object())


def test_column_tiles_of_a_one_gpu_sweep(tmp_path, monkeypatch):
    """SweepEngine(column_tiles=T): a sweep as T passes over T column ranges of the same tables (embedder.py:92 is
    independent per column) -- same embeddings, deltas and Embedder decisions as the plain sweep for even and ragged
    widths, with class rows; the default picks two tiles only where it was measured to win (rows that fill a wave,
    skewed reads, a table well beyond the Infinity Cache: profiles/r05_column_tiles_ab.md)."""
    from clane_amd import engine as E
    gold = load_golden("g5_symkarate_d16_g0.76.npz")
    k = load_golden("g2_karate_csr.npz")
    src, dst = (gold["edge_src"], gold["edge_dst"]) if "edge_src" in gold.files else (k["edge_src"], k["edge_dst"])
    vids = gold["vertex_ids"] if "vertex_ids" in gold.files else k["vertex_ids"]
    root = write_data_root(tmp_path / "g", vids, src, dst, gold["X"])
    for T in (2, 3):
        g = Graph(root)
        eng = SweepEngine(g.csr, g.X, "cpu", OracleKernels(), column_tiles=T, class_threshold=4, class_chunk=64)
        g._attach_engine(eng)
        assert len(eng.tiles) == T and eng.kernel_config()["column_tiles"] == T     # d = 16: four packs (2 + 2, 2 + 1 + 1)
        assert eng.launches_per_sweep() == len(eng.blocks) * len(eng.tiles)
        emb = Embedder(g, CosineSimilarity(), torch.device("cpu"), gamma=float(gold["gamma"]),
                       tolerence=int(gold["tolerence"]), verbose=False)
        emb.iterate()
        assert O.rel_l2(g.Z, torch.from_numpy(gold["Z_final"])) < 1e-6
    # ragged width, three tiles of 4 / 3 / 3 packs, deltas sweep by sweep
    rng = np.random.default_rng(5)
    V, d = 400, 40
    deg = rng.integers(0, 30, size=V)
    rowptr = np.zeros(V + 1, dtype=np.int64)
    np.cumsum(deg, out=rowptr[1:])
    cols = np.concatenate([np.sort(rng.choice(V, size=n, replace=False)) for n in deg]).astype(np.int32)
    csr = HostCSR(V, rowptr, cols)
    X = torch.from_numpy(rng.standard_normal((V, d)).astype(np.float32))
    one = SweepEngine(csr, X, "cpu", OracleKernels(), column_tiles=1, class_threshold=8, class_chunk=64)
    three = SweepEngine(csr, X, "cpu", OracleKernels(), column_tiles=3, class_threshold=8, class_chunk=64)
    assert three.tiles == [(0, 16), (16, 28), (28, 40)] and one.tiles == [(0, 40)]
    for eng in (one, three):
        eng.build_P()
    for _ in range(3):
        a, b = one.sweep(0.8), three.sweep(0.8)
        assert b == pytest.approx(a, rel=1e-6) and O.rel_l2(three.get_Z(), one.get_Z()) < 1e-6
    # tiles x class rows phased in time (profiles/r05_tiles_phases_ab.jsonl: four phases under two tiles is what K1,
    # which still reads whole rows, wants): the phases reorder a class row's edges, the tiles cut its columns
    for phases in (2, 4):
        phased = SweepEngine(csr, X, "cpu", OracleKernels(), column_tiles=3, class_threshold=8, class_chunk=64,
                             class_phases=phases, phase_threshold=16)
        assert phased.class_phases == phases and phased.kernel_config()["class_phases"] == phases
        plain = SweepEngine(csr, X, "cpu", OracleKernels(), column_tiles=1, class_threshold=8, class_chunk=64)
        for eng in (plain, phased):
            eng.build_P()
        assert O.rel_l2(phased.P_global(), plain.P_global()) < 1e-6
        for _ in range(3):
            a, b = plain.sweep(0.8), phased.sweep(0.8)
            assert b == pytest.approx(a, rel=1e-6) and O.rel_l2(phased.get_Z(), plain.get_Z()) < 1e-6
    with pytest.raises(ValueError, match="column_tiles"):
        SweepEngine(csr, X, "cpu", OracleKernels(), column_tiles=0)
    # the prepared default (xcd.PHASES_UNDER_COLUMN_TILES: 0 = by the tile's width) applies to tiled engines only
    from clane_amd import xcd
    monkeypatch.setattr(xcd, "PHASES_UNDER_COLUMN_TILES", 4)
    assert SweepEngine(csr, X, "cpu", OracleKernels(), column_tiles=3, class_threshold=8).class_phases == 4
    assert SweepEngine(csr, X, "cpu", OracleKernels(), column_tiles=1, class_threshold=8).class_phases == one.class_phases
    assert SweepEngine(csr, X, "cpu", OracleKernels(), column_tiles=3, class_threshold=8, class_phases=2).class_phases == 2
    monkeypatch.setattr(xcd, "PHASES_UNDER_COLUMN_TILES", 0)
    assert SweepEngine(csr, X, "cpu", OracleKernels(), column_tiles=3, class_threshold=8).class_phases == three.class_phases
    # the default: only where it pays (cache sizes scaled down 32x so that a 40 MB table stands in for a 1.3 GB one)
    from clane_amd import plan
    monkeypatch.setattr(E, "INFINITY_CACHE_BYTES", plan.INFINITY_CACHE_BYTES // 32)
    monkeypatch.setattr(plan, "L2_BYTES_ALL_XCDS", plan.L2_BYTES_ALL_XCDS // 32)
    V, E_ = 40_000, 640_000                      # 1-KiB rows; 16 edges a row
    rowptr = np.arange(0, E_ + 1, 16, dtype=np.int64)
    skewed = HostCSR(V, rowptr, rng.integers(0, 500, size=E_).astype(np.int32))
    even = HostCSR(V, rowptr, rng.integers(0, V, size=E_).astype(np.int32))
    Xz = torch.zeros(V, 256)
    assert len(SweepEngine(skewed, Xz, "cpu", OracleKernels()).tiles) == 2
    assert len(SweepEngine(even, Xz, "cpu", OracleKernels()).tiles) == 1                    # nothing to keep in a cache
    assert len(SweepEngine(skewed, Xz, "cpu", OracleKernels(), column_tiles=1).tiles) == 1  # obeyed
    assert len(SweepEngine(skewed, Xz[:, :128].contiguous(), "cpu", OracleKernels()).tiles) == 1   # 512-byte rows lose
    assert len(SweepEngine(skewed, Xz[:, :200].contiguous(), "cpu", OracleKernels()).tiles) == 1   # tiles would not be 128 columns
    assert len(SweepEngine(skewed, Xz.bfloat16(), "cpu", OracleKernels()).tiles) == 1              # measured for fp32 only
    wide = SweepEngine(skewed, torch.zeros(V, 512), "cpu", OracleKernels())
    assert wide.tiles == [(0, 128), (128, 256), (256, 384), (384, 512)]                             # tiles of 128 columns
    small = HostCSR(10_000, rowptr[:10_001], skewed.colidx[:160_000])
    assert len(SweepEngine(small, Xz[:10_000], "cpu", OracleKernels()).tiles) == 1            # fits the (scaled) Infinity Cache
    assert E.MIN_HOT_READ_SHARE == 0.2
