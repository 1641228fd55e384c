"""N > 1 path on CPU: `gloo` ranks run the SweepEngine in each of its multi-GPU divisions -- column split (the
default: no exchange per sweep, partial dot products all-reduced in build_P) and the row splits (halo table /
chunk-major in-place all-gather per chunk) -- with the oracle-backed test double as kernels, and every rank
must reproduce the single-process oracle."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, chunks, name, out_dir, exchange):
    sys.path.insert(0, str(ROOT))
    torch.set_num_threads(1)        # `world` processes share this box's few cores: no intra-op pools on top
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from clane_amd.embedder import Embedder
        from clane_amd.engine import SweepEngine
        from clane_amd.graph import Graph
        from clane_amd.similarity import CosineSimilarity
        from oracle import clane_oracle as O
        from tests.conftest import load_golden, write_data_root
        from tests.oracle_kernels import OracleKernels

        gold = load_golden(name)
        k = load_golden("g2_karate_csr.npz")
        src, dst = (gold["edge_src"], gold["edge_dst"]) if "edge_src" in gold.files else (k["edge_src"], k["edge_dst"])
        vids = gold["vertex_ids"] if "vertex_ids" in gold.files else k["vertex_ids"]
        root = write_data_root(Path(out_dir) / f"r{rank}", vids, src, dst, gold["X"])
        g = Graph(root)
        eng = SweepEngine(g.csr, g.X, "cpu", OracleKernels(), process_group=dist.group.WORLD, chunks=chunks, seed=3,
                          exchange=exchange)
        g._attach_engine(eng)
        assert eng.world == world and eng.halo == (exchange == "halo")
        full = (world - 1) * -(-len(g) // world) * eng.ld * 4
        if exchange == "allgather":                                        # live / quiet split
            n_quiet = int((~g.csr.live_mask()).sum())
            assert eng.part.quiet_per_rank == -(-n_quiet // world)
        if exchange == "columns":
            from clane_amd.engine import column_slice
            assert eng.columns and eng.exchange_bytes_per_sweep() == 0 and eng.part.n_local >= len(g)
            assert (eng.col0, eng.col1) == column_slice(g.X.shape[1], g.X.dtype, world, rank)
            widths = [None] * world
            dist.all_gather_object(widths, eng.d)
            assert sum(widths) == g.X.shape[1]                                # every column exactly once
        else:
            assert 0 <= eng.exchange_bytes_per_sweep() <= full + 4 * eng.ld * 4 * world
            received = [None] * world       # a rank may read nothing remote (few rows, all of them sinks): not all may
            dist.all_gather_object(received, eng.exchange_bytes_per_sweep())
            assert sum(received) > 0 and (world > 4 or received[rank] > 0)

        # build_P: every rank assembles the full P in the reference's (row, col) order
        P = g.build_P(CosineSimilarity())
        np.testing.assert_allclose(P.values().numpy(), gold["P0_values"], rtol=3e-6, atol=1e-7)

        # sweeps: full Z on every rank after the in-place all-gather, global delta after the all-reduce
        gamma = float(gold["gamma"])
        rowptr, colidx = g.csr.rowptr, g.csr.colidx
        X = torch.from_numpy(gold["X"])
        P_or = O.build_P_values(rowptr, colidx, X)
        Z = X.clone()
        for _ in range(4):
            delta = eng.sweep(gamma)
            Z, d_or = O.sweep(rowptr, colidx, P_or, X, Z, gamma)
            assert abs(delta - float(d_or)) <= 1e-5 * max(1.0, float(d_or))
            assert O.rel_l2(eng.get_Z(), Z) < 1e-6
        # both ping-pong buffers agree on rows without out-edges (CLANE_SPMM_SINKS_UNTOUCHED invariant)
        if not eng.halo:
            sink_pos = eng.pos[torch.from_numpy(np.diff(rowptr) == 0)]
            assert torch.equal(eng.Zbuf[0][sink_pos], eng.Zbuf[1][sink_pos])

        # full Embedder control flow on top: identical decisions on every rank, final Z = reference
        g2 = Graph(root)
        g2._attach_engine(SweepEngine(g2.csr, g2.X, "cpu", OracleKernels(), process_group=dist.group.WORLD,
                                      chunks=chunks, seed=3, exchange=exchange))
        emb = Embedder(g2, CosineSimilarity(), torch.device("cpu"), gamma=gamma, tolerence=int(gold["tolerence"]),
                       verbose=False)
        emb.iterate()
        assert O.rel_l2(g2.Z, torch.from_numpy(gold["Z_final"])) < 1e-6
        counts = [None] * world
        dist.all_gather_object(counts, (emb.sweep_counts, eng.estimated_sweep_seconds()))
        assert all(c == counts[0] for c in counts)      # same decisions, and the estimate behind the lagged check agrees

        # --save_history on several ranks: every rank stages only the part of Z it holds, the writer
        # threads hand the parts over through a directory, rank 0's sink gets whole matrices in order -- the same
        # ones a single process keeps in history["Z"]
        g3 = Graph(root)
        g3._attach_engine(SweepEngine(g3.csr, g3.X, "cpu", OracleKernels(), process_group=dist.group.WORLD,
                                      chunks=chunks, seed=3, exchange=exchange))
        got = []
        emb3 = Embedder(g3, CosineSimilarity(), torch.device("cpu"), gamma=gamma, tolerence=3, verbose=False,
                        save_history=True, history_sink=lambda o, s, Z: got.append((o, s, Z)),
                        history_parts_dir=Path(out_dir) / "parts")
        emb3.iterate()
        Z3 = g3.Z                                                       # collective: every rank takes part
        if rank == 0:
            g4 = Graph(root)
            g4._attach_engine(SweepEngine(g4.csr, g4.X, "cpu", OracleKernels()))
            solo = Embedder(g4, CosineSimilarity(), torch.device("cpu"), gamma=gamma, tolerence=3, verbose=False,
                            save_history=True)
            solo.iterate()
            want = [(o, s_, Z) for o, zs in enumerate(solo.history["Z"]) for s_, Z in enumerate(zs)]
            assert [(o, s_) for o, s_, _ in got][:8] == [(o, s_) for o, s_, _ in want][:8] and len(got) > 8
            for (_, _, Za), (_, _, Zb) in zip(got[:8], want[:8]):       # far from the fixed point: sweep for sweep
                assert O.rel_l2(Za, Zb) < 1e-6
            assert O.rel_l2(got[-1][2], Z3) < 1e-7                    # the last matrix handed over IS the result
        else:
            assert got == []                                            # only rank 0's sink sees matrices
        dist.barrier()
        if rank == 0:
            assert [p_ for p_ in (Path(out_dir) / "parts").rglob("*") if p_.is_file()] == []     # every part was consumed
        (Path(out_dir) / f"ok{rank}").write_text("ok")
        dist.barrier()              # leave together: a rank tearing gloo down while others still talk can abort
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["columns", "halo", "allgather", "allgather_all"])
@pytest.mark.parametrize("name,chunks", [("g5_symkarate_d16_g0.76.npz", 1), ("g5_symkarate_d16_g0.76.npz", 3),
                                         ("g4_karate_d2.npz", 2)])
def test_two_rank_gloo_matches_reference(tmp_path, name, chunks, exchange):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), chunks, name, str(tmp_path), exchange), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


@pytest.mark.parametrize("exchange", ["halo", "columns"])
def test_three_rank_gloo(tmp_path, exchange):
    """Odd world size: every (source, destination, chunk) list of the halo exchange is exercised; the column
    split is uneven (4 packs of d=16 over 3 ranks)."""
    world = 3
    mp.spawn(_worker, args=(world, _free_port(), 2, "g5_symkarate_d16_g0.5.npz", str(tmp_path), exchange), nprocs=world,
             join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


@pytest.mark.parametrize("exchange,name,chunks", [
    ("columns", "g5_symkarate_d16_g0.76.npz", 1),          # 4 packs of d=16: ranks 4-7 hold no column
    ("allgather_all", "g4_karate_d2.npz", 1),              # north_star's plan: 34 rows over 8 ranks, padded spans
    ("allgather_all", "g5_symkarate_d16_g0.76.npz", 4),    # 5 rows per rank in 4 chunks: 2-row and 1-row chunks
    ("allgather", "g4_karate_d2.npz", 2),                  # live / quiet split: karate has 9 sinks + never-read rows
    ("halo", "g4_karate_d2.npz", 1),
    ("halo", "g5_symkarate_d16_g0.76.npz", 2),
    ("halo", "g11_hubs320_d8_g0.9.npz", 4),                # 320 rows, hubs: 40 rows per rank, 10 per chunk
])
def test_eight_rank_gloo_every_division(tmp_path, exchange, name, chunks):
    """The driver's largest case, rehearsed on CPU with the world size it runs: 8 ranks in every division that
    `bench.py --gpus 8` executes (columns, then allgather_all = north_star's literal plan, then allgather / halo), on
    graphs with fewer rows than 8 x chunk x a comfortable chunk size -- uneven deals, padded in-place all-gather spans,
    ranks whose chunks hold one row or none.  Same collective sequence on every rank, same result as the reference."""
    world = 8
    mp.spawn(_worker, args=(world, _free_port(), chunks, name, str(tmp_path), exchange), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


@pytest.mark.parametrize("exchange", ["allgather_all", "allgather", "columns"])
def test_one_rank_group_with_forced_collectives(tmp_path, exchange):
    """TorchComm(force_collectives=True) over a ONE-rank group: the engine keeps the division it is given and issues
    every collective (each the identity) -- the rehearsal the GPU suite runs on RCCL with the box's single GPU."""
    sys.path.insert(0, str(ROOT))
    from clane_amd.comm import TorchComm
    from clane_amd.engine import SweepEngine
    from clane_amd.graph import Graph
    from oracle import clane_oracle as O
    from tests.conftest import load_golden, write_data_root
    from tests.oracle_kernels import OracleKernels
    gold, k = load_golden("g5_symkarate_d16_g0.76.npz"), load_golden("g2_karate_csr.npz")
    g = Graph(write_data_root(tmp_path / "g", k["vertex_ids"], gold["edge_src"], gold["edge_dst"], gold["X"]))
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1)
    try:
        comm = TorchComm(dist.group.WORLD, force_collectives=True)
        eng = SweepEngine(g.csr, g.X, "cpu", OracleKernels(), comm=comm, exchange=exchange, chunks=3, shuffle=False)
        assert eng.world == 1 and eng.exchange == exchange and eng.columns == (exchange == "columns")
        eng.build_P()
        np.testing.assert_allclose(eng.P_global().numpy(), gold["P0_values"], rtol=3e-6, atol=1e-7)
        X = torch.from_numpy(gold["X"])
        P_or = O.build_P_values(g.csr.rowptr, g.csr.colidx, X)
        Z = X.clone()
        for _ in range(3):
            delta = eng.sweep(float(gold["gamma"]))
            Z, d_or = O.sweep(g.csr.rowptr, g.csr.colidx, P_or, X, Z, float(gold["gamma"]))
            assert abs(delta - float(d_or)) <= 1e-5 * max(1.0, float(d_or))
        assert O.rel_l2(eng.get_Z(), Z) < 1e-6
        assert comm.calls["all_reduce"] >= 4 and (exchange == "columns" or comm.calls["all_gather"] == 9)
        plain = SweepEngine(g.csr, g.X, "cpu", OracleKernels(), comm=TorchComm(dist.group.WORLD), exchange=exchange)
        assert plain.exchange == "none" and not plain.columns           # without the flag: the one-GPU plan
    finally:
        dist.destroy_process_group()


def _uneven_columns_worker(rank: int, world: int, port: int, out_dir: str):
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(0)
        V, d = 500, 132                                  # 33 packs of 4 floats: 17 + 16 over two ranks
        deg = rng.integers(0, 12, size=V)
        deg[[3, 77, 200, 333]] = [100, 150, 200, 70]     # between the class thresholds of the two widths (64 / 256)
        rowptr = np.zeros(V + 1, dtype=np.int64)
        np.cumsum(deg, out=rowptr[1:])
        colidx = np.concatenate([np.sort(rng.choice(V, size=k, replace=False)) for k in deg]).astype(np.int32)
        from clane_amd.engine import SweepEngine, column_slice, lanes_per_row
        from clane_amd.partition import HostCSR
        from oracle import clane_oracle as O
        from tests.oracle_kernels import OracleKernels
        csr = HostCSR(V, rowptr, colidx)
        X = torch.from_numpy(rng.standard_normal((V, d)).astype(np.float32))
        widths = [column_slice(d, X.dtype, world, r) for r in range(world)]
        assert [lanes_per_row(c1 - c0, X.dtype) for c0, c1 in widths] == [32, 16]     # the ranks' kernels differ ...
        eng = SweepEngine(csr, X, "cpu", OracleKernels(), process_group=dist.group.WORLD, exchange="columns")
        everyone = [None] * world
        dist.all_gather_object(everyone, (eng.class_threshold, eng.class_phases, eng.phase_threshold, eng.class_affinity,
                                          eng.local.colidx.tobytes()))
        assert all(e == everyone[0] for e in everyone)   # ... their edge order must not: build_P all-reduces P element-wise
        eng.build_P()
        P_or = O.build_P_values(rowptr, colidx, X)
        assert O.rel_l2(eng.P_global(), P_or) < 3e-6
        Z = X.clone()
        for _ in range(3):
            delta = eng.sweep(0.76)
            Z, d_or = O.sweep(rowptr, colidx, P_or, X, Z, 0.76)
            assert abs(delta - float(d_or)) <= 1e-5 * max(1.0, float(d_or))
        assert O.rel_l2(eng.get_Z(), Z) < 1e-6
        (Path(out_dir) / f"ok{rank}").write_text("ok")
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_column_ranks_of_different_width_share_one_edge_order(tmp_path):
    """A column split whose slices straddle a lane-layout boundary (33 packs over 2 ranks: 32 and 16 lanes per row): the
    row-binning thresholds follow the row width, but the class-sorted edge order must be the same on every rank --
    build_P all-reduces the partial dot products element by element."""
    mp.spawn(_uneven_columns_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(2))


class _ScaledDot:
    """A plug-in similarity in the reference's protocol (similarity.py:5-7: a callable on [B, d] batches) whose score
    for a pair does not depend on the rest of the batch."""
    batchwise = True

    def is_trainable(self):
        return False

    def __call__(self, v1, v2):
        return (v1 * v2).sum(-1) / (1.0 + v1.pow(2).sum(-1).sqrt() * v2.pow(2).sum(-1).sqrt())


def _plugin_worker(rank: int, world: int, port: int, out_dir: str, exchange: str):
    sys.path.insert(0, str(ROOT))
    torch.set_num_threads(1)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from clane_amd.embedder import Embedder
        from clane_amd.engine import SweepEngine
        from clane_amd.graph import Graph
        from clane_amd.similarity import CosineSimilarity
        from oracle import clane_oracle as O
        from tests.conftest import load_golden, write_data_root
        from tests.oracle_kernels import OracleKernels
        gold, k = load_golden("g5_symkarate_d16_g0.76.npz"), load_golden("g2_karate_csr.npz")
        root = write_data_root(Path(out_dir) / f"r{rank}", k["vertex_ids"], gold["edge_src"], gold["edge_dst"], gold["X"])
        g = Graph(root)
        g._attach_engine(SweepEngine(g.csr, g.X, "cpu", OracleKernels(), process_group=dist.group.WORLD, chunks=2, seed=3,
                                     exchange=exchange))
        sim = _ScaledDot()
        if exchange == "columns":           # a plug-in needs whole rows
            with pytest.raises(NotImplementedError, match="whole rows"):
                g.build_P(sim)
        else:
            X = torch.from_numpy(gold["X"])
            rowptr, colidx = g.csr.rowptr, g.csr.colidx
            rows = torch.from_numpy(np.repeat(np.arange(len(g)), np.diff(rowptr)))
            want = O.segment_softmax(rowptr, sim(X[rows], X[torch.from_numpy(colidx).long()]))
            P = g.build_P(sim)              # every rank scores its own rows' edges; the result is the whole P everywhere
            np.testing.assert_allclose(P.values().numpy(), want.numpy(), rtol=1e-5, atol=1e-7)
            with pytest.raises(NotImplementedError, match="batchwise"):      # a batch-global measure: one GPU only
                g.build_P(lambda a, b: (a * b).sum(-1) / a.norm())
            # and the whole loop on top of it: same embeddings as one process with the same plug-in
            emb = Embedder(g, sim, torch.device("cpu"), gamma=0.5, tolerence=3, verbose=False)
            emb.iterate()
            Z = g.Z
            if rank == 0:
                g1 = Graph(root)
                g1._attach_engine(SweepEngine(g1.csr, g1.X, "cpu", OracleKernels()))
                Embedder(g1, sim, torch.device("cpu"), gamma=0.5, tolerence=3, verbose=False).iterate()
                assert O.rel_l2(Z, g1.Z) < 1e-6
        (Path(out_dir) / f"ok{rank}").write_text("ok")
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["halo", "allgather_all", "allgather", "columns"])
def test_plugin_similarity_on_two_ranks(tmp_path, exchange):
    """The reference's plug-in surface (__main__.py:39-48, graph.py:120-121) on several GPUs: with the rows divided,
    every rank calls a `batchwise` plug-in on the edges of its own rows; batch-global callables and column-divided
    engines are refused with a message that says what to do."""
    mp.spawn(_plugin_worker, args=(2, _free_port(), str(tmp_path), exchange), nprocs=2, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(2))
