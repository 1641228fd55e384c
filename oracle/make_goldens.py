"""Generate tests/golden/*.npz by running the REAL reference on fixed inputs.

TEST INFRASTRUCTURE.  Runs only in the build container, where the reference is
mounted read-only at /root/reference; it never travels to the GPU box.  Only
the resulting data (inputs + the reference's outputs) is committed.

    python oracle/make_goldens.py            # rewrites tests/golden/
    CLANE_GOLDEN_OUT=/tmp/g python oracle/make_goldens.py    # elsewhere, e.g. to compare with the committed fixtures

The reference needs one in-process alias to import under NumPy 2
(``from numpy import Inf`` at clane/embedder.py:1): ``numpy.Inf = numpy.inf``.
"""
from __future__ import annotations

import contextlib
import io
import os
import shutil
import sys
import tempfile
from pathlib import Path

import numpy as np

REF = Path("/root/reference")
OUT = Path(os.environ.get("CLANE_GOLDEN_OUT") or Path(__file__).resolve().parent.parent / "tests" / "golden")


def _import_reference():
    np.Inf = np.inf  # noqa: NPY201 -- alias the reference needs (embedder.py:1)
    sys.dont_write_bytecode = True
    sys.path.insert(0, str(REF))
    import clane.embedder as E
    import clane.graph as G
    import clane.similarity as S
    import clane.__main__ as M
    E.tqdm = lambda x, **k: x  # silence the per-vertex progress bar
    return G, S, E, M


def _write_root(root: Path, vertex_ids, edge_lines, C=None):
    root.mkdir(parents=True, exist_ok=True)
    (root / "V").write_text("\n".join(vertex_ids) + "\n")
    (root / "E").write_text("\n".join(f"{s}\t{d}" for s, d in edge_lines) + "\n")
    if C is not None:
        np.save(root / "C.npy", C)


def _run_embedder(G, S, E, root: Path, gamma: float, tol: int = 10):
    """Full reference iterate(); returns dict of captured outputs."""
    import torch
    g = G.Graph(root)
    A = g.A
    P0 = g.build_P(S.CosineSimilarity())
    emb = E.Embedder(g, S.CosineSimilarity(), torch.device("cpu"), gamma=gamma,
                     tolerence=tol, save_history=True)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        emb.iterate()
    sweep_counts = [len(h) for h in emb.history["Z"]]
    return dict(
        X=g.X.numpy(),
        A_indices=A.indices().numpy(),
        P0_values=P0.values().numpy(),
        Z_sweep1=emb.history["Z"][0][0].numpy(),
        Z_prop1=emb.history["Z"][0][-1].numpy(),
        Z_final=g.Z.numpy(),
        sweep_counts=np.array(sweep_counts),
        gamma=np.float64(gamma), tolerence=np.int64(tol),
    )


def karate_files():
    root = REF / "tests" / "data_root"
    vids = (root / "V").read_text().strip().split("\n")
    edges = [tuple(l.split("\t")) for l in (root / "E").read_text().strip().split("\n")]
    return vids, edges


def g10_labels_and_log(G, S, E):
    """G10: the reference's karate label file (tests/data_root/Y, a data file no code reads) and the per-sweep
    lines its propagate prints (embedder.py:104) on the G4 d=2 input -- pins the log format `tensor(25.7074) 10`."""
    import torch
    gold = np.load(OUT / "g4_karate_d2.npz")
    vids, edges = karate_files()
    tmp = Path(tempfile.mkdtemp(prefix="clane_gold_"))
    _write_root(tmp / "k", vids, edges, gold["X"])
    g = G.Graph(tmp / "k")
    emb = E.Embedder(g, S.CosineSimilarity(), torch.device("cpu"), gamma=float(gold["gamma"]),
                     tolerence=int(gold["tolerence"]))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        emb.propagate()
    y_lines = (REF / "tests" / "data_root" / "Y").read_text().strip().split("\n")
    np.savez_compressed(OUT / "g10_karate_labels_log.npz",
                        Y_ids=np.array([l.split("\t")[0] for l in y_lines]),
                        Y_classes=np.array([l.split("\t")[1] for l in y_lines]),
                        propagate_stdout=np.array(buf.getvalue().splitlines()))
    shutil.rmtree(tmp)


def main():
    import torch
    G, S, E, M = _import_reference()
    OUT.mkdir(parents=True, exist_ok=True)
    if sys.argv[1:] == ["g10"]:          # add G10 without re-drawing G9's unseeded content embeddings
        g10_labels_and_log(G, S, E)
        return
    tmp = Path(tempfile.mkdtemp(prefix="clane_gold_"))
    cs = S.CosineSimilarity()

    # ---- G1: CosineSimilarity known answers + batched (global-norm) behaviour
    g1 = torch.Generator().manual_seed(11)
    a4, b4 = torch.rand(4, 16, generator=g1), torch.rand(4, 16, generator=g1)
    a64, b64 = torch.rand(5, 7, generator=g1, dtype=torch.float64), torch.rand(5, 7, generator=g1, dtype=torch.float64)
    v = torch.tensor([1.0, 2.0, 3.0])
    np.savez_compressed(
        OUT / "g1_cosine.npz",
        same=cs(v, v).numpy(), orth=cs(torch.tensor([0.0, 1.0]), torch.tensor([1.0, 0.0])).numpy(),
        opp=cs(v, -v).numpy(),
        a4=a4.numpy(), b4=b4.numpy(), out4=cs(a4, b4).numpy(),
        a64=a64.numpy(), b64=b64.numpy(), out64=cs(a64, b64).numpy(),
        zeros_ones=cs(torch.zeros(3), torch.ones(3)).numpy(),
    )

    # ---- G2/G3/G4: karate (34 / 78), d=2 and d=16
    vids, edges = karate_files()
    sym_edges = sorted(set(edges) | {(d, s) for s, d in edges}, key=lambda e: (int(e[0]), int(e[1])))
    gk = G.Graph(REF / "tests" / "data_root", embedding_dim=2)
    A = gk.A
    np.savez_compressed(
        OUT / "g2_karate_csr.npz",
        vertex_ids=np.array(vids), edge_src=np.array([e[0] for e in edges]), edge_dst=np.array([e[1] for e in edges]),
        A_indices=A.indices().numpy(), A_values=A.values().numpy(),
        nbrs33=gk.get_nbrs(33).numpy(),
        nbrs=np.array([gk.get_nbrs(i).numpy() for i in range(34)], dtype=object),
        num_E=np.int64(len(gk.E)),
    )
    for d, seed, gamma in [(2, 0, 0.76), (16, 1, 0.74)]:
        X = torch.normal(0, 1, [34, d], generator=torch.Generator().manual_seed(seed)).numpy()
        root = tmp / f"karate_d{d}"
        _write_root(root, vids, edges, X)
        res = _run_embedder(G, S, E, root, gamma)
        np.savez_compressed(OUT / f"g4_karate_d{d}.npz", **res)

    # ---- G5: symmetrised karate (cyclic), d=16, gamma 0.76 / 0.5 ; d=2 gamma 0.76
    for d, seed, gamma in [(16, 2, 0.76), (16, 2, 0.5), (2, 3, 0.76)]:
        X = torch.normal(0, 1, [34, d], generator=torch.Generator().manual_seed(seed)).numpy()
        root = tmp / f"symk_d{d}_{gamma}"
        _write_root(root, vids, sym_edges, X)
        res = _run_embedder(G, S, E, root, gamma)
        res["edge_src"] = np.array([e[0] for e in sym_edges])
        res["edge_dst"] = np.array([e[1] for e in sym_edges])
        np.savez_compressed(OUT / f"g5_symkarate_d{d}_g{gamma}.npz", **res)

    # ---- G6: 4-node graph: duplicate edge, self-loop, sink, non-numeric ids, float64 C.npy
    vids6 = ["a", "b", "c", "d"]
    edges6 = [("a", "b"), ("a", "c"), ("a", "b"), ("b", "a"), ("d", "d")]
    X6 = torch.normal(0, 1, [4, 3], generator=torch.Generator().manual_seed(6), dtype=torch.float64).numpy()
    root = tmp / "g6"
    _write_root(root, vids6, edges6, X6)
    g6 = G.Graph(root, embedding_dim=128)  # embedding_dim ignored for shape, kept in .d
    res = _run_embedder(G, S, E, root, 0.76)
    res.update(vertex_ids=np.array(vids6), edge_src=np.array([e[0] for e in edges6]),
               edge_dst=np.array([e[1] for e in edges6]), num_E=np.int64(len(g6.E)),
               d_attr=np.int64(g6.d), P0_dense=g6.build_P(cs).to_dense().numpy())
    np.savez_compressed(OUT / "g6_tiny_f64.npz", **res)

    # ---- G7: README 5-node graph (README.md:18-33), d=2
    vids7 = ["1", "2", "3", "4", "5"]
    edges7 = [("1", "2"), ("1", "4"), ("2", "5"), ("3", "1")]
    X7 = torch.normal(0, 1, [5, 2], generator=torch.Generator().manual_seed(7)).numpy()
    root = tmp / "g7"
    _write_root(root, vids7, edges7, X7)
    res = _run_embedder(G, S, E, root, 0.76)
    res.update(vertex_ids=np.array(vids7), edge_src=np.array([e[0] for e in edges7]),
               edge_dst=np.array([e[1] for e in edges7]))
    np.savez_compressed(OUT / "g7_readme5.npz", **res)

    # ---- G11: a 320-vertex graph with what the karate-sized fixtures lack -- a hub row of 300 edges (above the
    # class threshold of narrow rows: the XCD-affine pass of the GPU path gets a fixture made by the real reference),
    # one of 100, self-loops, duplicate edge lines, string ids in shuffled order, sinks; d = 8, gamma = 0.9
    rng11 = np.random.default_rng(11)
    V11 = 320
    vids11 = [f"n{int(i):03d}" for i in rng11.permutation(V11)]
    edges11 = []
    for v in range(V11):
        if v in (5, 77, 150, 319):                       # sinks
            continue
        k = 300 if v == 17 else 100 if v == 200 else int(rng11.integers(1, 7))
        for u in rng11.choice(V11, size=k, replace=False):
            edges11.append((vids11[v], vids11[int(u)]))
    edges11 += [(vids11[3], vids11[3]), (vids11[17], vids11[17]), edges11[0], edges11[10], edges11[10]]   # self-loops, duplicates
    order11 = rng11.permutation(len(edges11))
    edges11 = [edges11[int(i)] for i in order11]         # arbitrary line order in the E file
    X11 = torch.normal(0, 1, [V11, 8], generator=torch.Generator().manual_seed(11)).numpy()
    root = tmp / "g11"
    _write_root(root, vids11, edges11, X11)
    res = _run_embedder(G, S, E, root, 0.9)
    res.update(vertex_ids=np.array(vids11), edge_src=np.array([e[0] for e in edges11]),
               edge_dst=np.array([e[1] for e in edges11]))
    np.savez_compressed(OUT / "g11_hubs320_d8_g0.9.npz", **res)

    # ---- G12: wide rows (d = 256: one 1-KiB row per wave instruction on the GPU, class threshold 64) with hub rows of
    # 120 and 70 edges and a cycle-rich background, gamma = 0.76
    rng12 = np.random.default_rng(12)
    V12 = 150
    vids12 = [str(1000 + i) for i in range(V12)]
    edges12 = []
    for v in range(V12):
        if v % 37 == 0:                                  # sinks
            continue
        k = 120 if v == 9 else 70 if v == 101 else int(rng12.integers(1, 9))
        for u in rng12.choice(V12, size=k, replace=False):
            edges12.append((vids12[v], vids12[int(u)]))
    X12 = torch.normal(0, 1, [V12, 256], generator=torch.Generator().manual_seed(12)).numpy()
    root = tmp / "g12"
    _write_root(root, vids12, edges12, X12)
    res = _run_embedder(G, S, E, root, 0.76)
    res.update(vertex_ids=np.array(vids12), edge_src=np.array([e[0] for e in edges12]),
               edge_dst=np.array([e[1] for e in edges12]))
    np.savez_compressed(OUT / "g12_hubs150_d256_g0.76.npz", **res)

    # ---- G8: Cora-shaped synthetic (2708 / 5429 / d=1433 binary BoW); literal loop is
    # 16 s/sweep so: build_P + ONE literal sweep through the reference's own propagate body.
    rng = np.random.default_rng(0)
    Vn, En, dn = 2708, 5429, 1433
    keys = rng.choice(Vn * Vn, size=En, replace=False)
    src8, dst8 = np.sort(keys) // Vn, np.sort(keys) % Vn
    X8 = (rng.random((Vn, dn)) < 18.0 / dn).astype(np.float32)
    X8[X8.sum(1) == 0, 0] = 1.0
    vids8 = [str(i) for i in range(Vn)]
    root = tmp / "g8"
    _write_root(root, vids8, [(str(s), str(d)) for s, d in zip(src8, dst8)], X8)
    g8 = G.Graph(root)
    P8 = g8.build_P(cs)
    emb8 = E.Embedder(g8, cs, torch.device("cpu"), gamma=0.76, tolerence=10)
    # one literal sweep = the body of embedder.py:84-94, driven through the reference objects
    cur = g8.Z.clone()
    A8 = g8.A  # hoisted: get_nbrs rebuilds A per call (O(E)); identical values
    crow = A8.indices()
    for idx_v, vtx in enumerate(g8.V):
        nb = crow[1][crow[0] == vtx.idx]
        if nb.size(0) == 0:
            continue
        w = P8[idx_v].coalesce().values().view(1, -1)
        vtx.z = (vtx.x + emb8.gamma * (w.mm(cur[nb]))).squeeze(0)
    Z1 = g8.Z
    delta1 = (Z1 - cur).absolute().sum()
    ones = np.nonzero(X8)
    np.savez_compressed(
        OUT / "g8_corashape.npz",
        V=np.int64(Vn), d=np.int64(dn), src=src8.astype(np.int32), dst=dst8.astype(np.int32),
        X_nz_row=ones[0].astype(np.int16), X_nz_col=ones[1].astype(np.int16),
        P_values=P8.values().numpy(), A_indices=P8.indices().numpy().astype(np.int32),
        Z1_head=Z1[:24].numpy(), Z1_rownorm=Z1.norm(dim=1).numpy(), Z1_rowsum=Z1.sum(1).numpy(),
        delta1=delta1.numpy(), gamma=np.float64(0.76),
    )

    # ---- G9: CLI run of the reference's own tests/config.yaml with --save_history
    cwd = os.getcwd()
    work = tmp / "g9"
    work.mkdir()
    os.symlink(REF / "tests", work / "tests")
    os.chdir(work)
    try:
        args = M.get_parser().parse_args(["--data_root", "./tests/data_root", "--output_root", "./test_output",
                                          "--config_file", "./tests/config.yaml", "--save_history"])
        with contextlib.redirect_stdout(io.StringIO()) as so:
            M.embedding(args)
        listing = sorted(str(p.relative_to(work / "test_output")) for p in (work / "test_output").rglob("*.npy"))
        shapes = [np.load(work / "test_output" / p).shape for p in listing]
        banner = [l for l in so.getvalue().splitlines() if not l.startswith("tensor(")]
    finally:
        os.chdir(cwd)
    np.savez_compressed(OUT / "g9_cli.npz", listing=np.array(listing), shapes=np.array(shapes),
                        banner=np.array(banner),
                        parser_dests=np.array(sorted(a.dest for a in M.get_parser()._actions)))

    shutil.rmtree(tmp)
    g10_labels_and_log(G, S, E)
    for p in sorted(OUT.glob("*.npz")):
        print(f"{p.name:36s} {p.stat().st_size:9d} B")


if __name__ == "__main__":
    main()
