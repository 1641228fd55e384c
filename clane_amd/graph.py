"""Graph -- the reference's ``clane/graph.py`` surface over CSR arrays and GPU-resident state.

Same constructor, attributes and methods as the reference (``Graph(data_root, embedding_dim)``,
``d, vertex_ids, X, V, E, dispense_pair, A, get_nbrs, build_P, Z, set_Z, len()``), same files
(``V``: ids one per line; ``E``: ``src\\tdst`` per line; optional ``C.npy`` / ``C.pt``), same
exceptions.  What differs is the representation: one O(|E|) parse into CSR instead of a Python
``Edge`` object per line and an O(|V|) ``list.index`` per endpoint (graph.py:79-89), and the
embeddings live in HBM inside a ``SweepEngine`` instead of one tensor per ``Vertex``.
"""
from __future__ import annotations

from pathlib import Path
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _hip
from .partition import HostCSR


# ---- file parsing (graph.py:43-47, 72-81) ---------------------------------------------------
def read_vertex_ids(data_root: Path) -> List[str]:
    with open(Path(data_root).joinpath("V"), "r") as io:          # FileNotFoundError propagates
        return io.read().strip().split("\n")


def read_edge_indices(data_root: Path, vertex_ids: Sequence[str]):
    """-> (src, dst) int64 arrays, one entry per line of ``E`` (duplicates kept).

    An id resolves to the index of its FIRST occurrence in ``V`` (``list.index``,
    graph.py:81); an unknown id or a line without exactly one tab raises ValueError.
    Parsed by the native loader (csrc/host_loader.cpp, O(|V| + |E|)) when libclane_host.so is
    built, else by the equivalent Python loop below.
    """
    data_root = Path(data_root)
    with open(data_root.joinpath("E"), "r"):                    # FileNotFoundError propagates, as upstream
        pass
    native = _native_parse_edges(data_root)
    if native is not None:
        return native
    return _python_parse_edges(data_root, vertex_ids)


def _python_parse_edges(data_root: Path, vertex_ids: Sequence[str]):
    with open(Path(data_root).joinpath("E"), "r") as io:
        lines = io.read().strip().split("\n")
    first = {}
    for i, vid in enumerate(vertex_ids):
        first.setdefault(vid, i)
    src = np.empty(len(lines), dtype=np.int64)
    dst = np.empty(len(lines), dtype=np.int64)
    for k, line in enumerate(lines):
        parts = line.split("\t")
        if len(parts) != 2:
            raise ValueError(f"E line {k + 1}: expected 'src\\tdst', got {line!r}")
        try:
            src[k], dst[k] = first[parts[0]], first[parts[1]]
        except KeyError as exc:
            raise ValueError(f"{exc.args[0]!r} is not in list") from None
    return src, dst


_HOST_LIB = None


def _host_lib():
    """ctypes handle of libclane_host.so (host-side parser), or None when it has not been built."""
    global _HOST_LIB
    if _HOST_LIB is None:
        import ctypes as C
        path = Path(__file__).resolve().parent / "libclane_host.so"
        if not path.exists():
            _HOST_LIB = False
        else:
            lib = C.CDLL(str(path))
            lib.clane_count_lines.restype = C.c_int64
            lib.clane_count_lines.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
            lib.clane_parse_edges.restype = C.c_int64
            lib.clane_parse_edges.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_char_p,
                                              C.c_int]
            _HOST_LIB = lib
    return _HOST_LIB or None


def _native_parse_edges(data_root: Path):
    lib = _host_lib()
    if lib is None:
        return None
    import ctypes as C
    err = C.create_string_buffer(256)
    e_path, v_path = str(data_root / "E").encode(), str(data_root / "V").encode()
    n = lib.clane_count_lines(e_path, err, len(err))
    if n < 0:
        raise OSError(err.value.decode())
    src = np.empty(n, dtype=np.int64)
    dst = np.empty(n, dtype=np.int64)
    got = lib.clane_parse_edges(v_path, e_path, src.ctypes.data, dst.ctypes.data, n, err, len(err))
    if got == -4:           # not valid UTF-8: the Python loop opens the files in text mode and raises what upstream raises
        return None
    if got in (-2, -3):
        raise ValueError(err.value.decode(errors="replace"))
    if got < 0:
        raise OSError(err.value.decode())
    return src[:got], dst[:got]


def _cached_edge_indices(data_root: Path, vertex_ids: Sequence[str]):
    """read_edge_indices() through a small binary cache beside the text files."""
    stamp = np.array([[p.stat().st_size, p.stat().st_mtime_ns] for p in (data_root / "V", data_root / "E")],
                     dtype=np.int64)
    path = data_root / ".clane_edges.npz"
    if path.exists():
        try:
            with np.load(path) as z:
                if np.array_equal(z["stamp"], stamp):
                    return z["src"], z["dst"]
        except Exception:
            pass                                            # unreadable / stale cache: parse again
    src, dst = read_edge_indices(data_root, vertex_ids)
    try:
        np.savez(path, stamp=stamp, src=src, dst=dst)
    except OSError:
        pass                                                # read-only data_root: no cache, same result
    return src, dst


def csr_from_edges(num_vertices: int, src: np.ndarray, dst: np.ndarray, device="auto") -> HostCSR:
    """Coalesced adjacency (graph.py:104-110): sorted by (src, dst), duplicates merged, self-loops kept.
    One sort of the (src, dst) keys.  Large edge lists are sorted on a GPU (40M edges: 5.4 s with numpy, 2.4 s with
    torch on 8 host cores, 0.2 s on the card including both copies); a host utility either way.  ``device``: the card
    to borrow -- "auto" (the current one, when there is one: the embedding loop needs it anyway), a device, or None /
    "cpu" for the host.  A card without room for the keys and the sort's scratch (several times 8 bytes per edge)
    is not an error: the host sorts instead."""
    n = int(num_vertices)
    key = (torch.from_numpy(np.ascontiguousarray(src, dtype=np.int64)) * n
           + torch.from_numpy(np.ascontiguousarray(dst, dtype=np.int64)))
    if device == "auto":
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
    on_card = (device is not None and torch.device(device).type == "cuda" and key.numel() >= GPU_SORT_MIN_EDGES)
    if on_card:
        try:
            return _csr_from_keys(n, _sorted_unique(key.to(device), 0, n * n))
        except torch.cuda.OutOfMemoryError:
            torch.cuda.empty_cache()
    return _csr_from_keys(n, _sorted_unique(key, 0, n * n))


def _csr_from_keys(n: int, key: torch.Tensor) -> HostCSR:
    rows = torch.div(key, n, rounding_mode="floor")
    counts = torch.bincount(rows, minlength=n).cpu().numpy()
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(counts, out=rowptr[1:])
    return HostCSR(n, rowptr, (key - rows * n).to(torch.int32).cpu().numpy())


def _sorted_unique(key: torch.Tensor, lo: int, hi: int) -> torch.Tensor:
    """torch.unique (sorted) of int64 keys in [lo, hi); lists beyond what one torch sort takes (2^31 - 1 elements)
    are split at the middle of the key range and the halves done one after the other."""
    if key.numel() <= SORT_MAX_ELEMENTS or hi - lo < 2:
        return torch.unique(key)
    mid = (lo + hi) // 2
    low = key < mid
    return torch.cat([_sorted_unique(key[low], lo, mid), _sorted_unique(key[~low], mid, hi)])


GPU_SORT_MIN_EDGES = 1 << 20
SORT_MAX_ELEMENTS = (1 << 31) - 1


def _parse_dtype(dtype) -> torch.dtype:
    if isinstance(dtype, torch.dtype):
        return dtype
    names = {"float32": torch.float32, "fp32": torch.float32, "f32": torch.float32,
             "float64": torch.float64, "fp64": torch.float64, "f64": torch.float64, "double": torch.float64,
             "bfloat16": torch.bfloat16, "bf16": torch.bfloat16}
    try:
        return names[str(dtype).lower().replace("torch.", "")]
    except KeyError:
        raise ValueError(f"unsupported embedding dtype {dtype!r} (float32, float64 or bfloat16)") from None


# ---- object-model facade ----------------------------------------------------------------------
class Vertex(object):
    """View of one vertex (reference graph.py:9-21).  ``x`` / ``z`` are rows of the graph's matrices."""
    __slots__ = ("_g", "idx", "id_")

    def __init__(self, graph: "Graph", idx: int, id_) -> None:
        self._g, self.idx, self.id_ = graph, idx, id_

    @property
    def x(self) -> torch.Tensor:
        return self._g.X[self.idx]

    @property
    def z(self) -> torch.Tensor:
        return self._g._host_Z()[self.idx]

    @z.setter
    def z(self, value: torch.Tensor) -> None:
        self._g._set_row(self.idx, value)

    @property
    def outgoing_indices(self) -> List[int]:
        return self._g._raw_neighbours(self.idx, out=True)

    @property
    def incoming_indices(self) -> List[int]:
        return self._g._raw_neighbours(self.idx, out=False)


class Edge(object):
    """reference graph.py:24-30."""
    __slots__ = ("src", "dst")

    def __init__(self, src: Vertex, dst: Vertex) -> None:
        self.src, self.dst = src, dst


class _LazySeq:
    """len()/index/iterate without materialising one Python object per element."""

    def __init__(self, n: int, make):
        self._n, self._make = n, make

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._make(j) for j in range(*i.indices(self._n))]
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        return self._make(i)

    def __iter__(self):
        return (self._make(i) for i in range(self._n))


class Graph(torch.utils.data.Dataset):
    # plug-in similarity callables (build_P): edge batches of a `batchwise` callable are gathered this many bytes
    # at a time; any other callable gets the reference's single call, refused above this size (None: 80 % of the
    # free HBM at the time of the call)
    PLUGIN_CHUNK_BYTES = 1 << 30
    PLUGIN_SINGLE_CALL_MAX_BYTES = None
    PLUGIN_EXCHANGE = "halo"        # division of an engine first created for a plug-in similarity (needs whole rows)

    def __init__(self, data_root: Path, embedding_dim: int = 128, dtype=None, cache: bool = False) -> None:
        """``dtype`` (extension, default None = keep what the files hold, as upstream): storage type of the
        embeddings on the GPU -- "float32", "float64" or "bfloat16" (bf16 storage, fp32 accumulate and P).
        ``cache`` (extension, default off): keep the parsed edge list in ``data_root/.clane_edges.npz`` and reuse
        it while ``V`` and ``E`` are unchanged (size + mtime), so a 40M-edge graph is parsed once."""
        super().__init__()
        data_root = Path(data_root)
        self.d = embedding_dim
        self.vertex_ids = read_vertex_ids(data_root)

        # Content embeddings C: C.npy -> C.pt -> N(0,1) (graph.py:50-58).  dtype is preserved.
        try:
            self.X = torch.from_numpy(np.load(data_root.joinpath("C.npy")))
        except FileNotFoundError:
            try:
                self.X = torch.load(data_root.joinpath("C.pt"))
            except FileNotFoundError:
                self.X = torch.normal(0, 1, [len(self.vertex_ids), self.d])
        if dtype is not None:
            self.X = self.X.to(_parse_dtype(dtype))
        if self.X.dim() != 2 or self.X.shape[0] != len(self.vertex_ids):
            raise ValueError(f"content embeddings have shape {tuple(self.X.shape)}, expected "
                             f"[{len(self.vertex_ids)}, d]")

        self._raw_src, self._raw_dst = (_cached_edge_indices(data_root, self.vertex_ids) if cache
                                        else read_edge_indices(data_root, self.vertex_ids))
        self.csr = csr_from_edges(len(self.vertex_ids), self._raw_src, self._raw_dst)

        self.V = _LazySeq(len(self.vertex_ids), lambda i: Vertex(self, i, self.vertex_ids[i]))
        self.E = _LazySeq(len(self._raw_src),
                          lambda k: Edge(self.V[int(self._raw_src[k])], self.V[int(self._raw_dst[k])]))
        self.dispense_pair = False

        self._Z_host: Optional[torch.Tensor] = None   # authoritative only while no engine exists / when dirty
        self._dirty = False
        self._engine = None
        self._raw_order = {}

    @classmethod
    def from_csr(cls, csr: HostCSR, X: torch.Tensor, vertex_ids=None) -> "Graph":
        """A Graph over an adjacency that already is a CSR in memory (synthetic inputs of bench.py and the
        tests; extension -- upstream only reads ``data_root``).  ``E`` then lists the CSR's edges."""
        if X.dim() != 2 or X.shape[0] != csr.num_vertices:
            raise ValueError(f"content embeddings have shape {tuple(X.shape)}, expected [{csr.num_vertices}, d]")
        g = cls.__new__(cls)
        torch.utils.data.Dataset.__init__(g)
        n = csr.num_vertices
        g.d, g.X, g.csr = int(X.shape[1]), X, csr
        g.vertex_ids = vertex_ids if vertex_ids is not None else _LazySeq(n, str)
        g._raw_src = np.repeat(np.arange(n, dtype=np.int64), csr.outdeg())
        g._raw_dst = csr.colidx.astype(np.int64)
        g.V = _LazySeq(n, lambda i: Vertex(g, i, g.vertex_ids[i]))
        g.E = _LazySeq(csr.num_edges, lambda k: Edge(g.V[int(g._raw_src[k])], g.V[int(g._raw_dst[k])]))
        g.dispense_pair = False
        g._Z_host, g._dirty, g._engine, g._raw_order = None, False, None, {}
        return g

    # ---- Dataset protocol ---------------------------------------------------------------
    def __len__(self):
        return len(self.vertex_ids)

    def __getitem__(self, idx):
        if self.dispense_pair:
            raise NotImplementedError("negative-pair sampling serves only the reference's trainable-similarity "
                                      "path (IterativeEmbedder), which is outside this package's scope")
        return idx

    # ---- adjacency (graph.py:104-116) -----------------------------------------------------
    def _edge_index(self) -> torch.Tensor:
        rows = np.repeat(np.arange(len(self), dtype=np.int64), self.csr.outdeg())
        return torch.from_numpy(np.stack([rows, self.csr.colidx.astype(np.int64)]))

    @property
    def A(self) -> torch.Tensor:
        idx = self._edge_index()
        return torch.sparse_coo_tensor(idx, torch.ones(idx.shape[1]), size=(len(self), len(self)),
                                       is_coalesced=True)

    def get_nbrs(self, idx: int) -> torch.LongTensor:
        a, b = self.csr.rowptr[idx], self.csr.rowptr[idx + 1]
        return torch.from_numpy(self.csr.colidx[a:b].astype(np.int64))

    def _raw_neighbours(self, idx: int, out: bool) -> List[int]:
        key, val = (self._raw_src, self._raw_dst) if out else (self._raw_dst, self._raw_src)
        if out not in self._raw_order:
            order = np.argsort(key, kind="stable")
            ptr = np.zeros(len(self) + 1, dtype=np.int64)
            np.cumsum(np.bincount(key, minlength=len(self)), out=ptr[1:])
            self._raw_order[out] = (order, ptr)
        order, ptr = self._raw_order[out]
        return val[order[ptr[idx]:ptr[idx + 1]]].tolist()

    # ---- GPU engine ---------------------------------------------------------------------
    def engine(self, device=None, cosine_mode: str = "reference", **engine_kwargs):
        """The SweepEngine holding this graph's state in HBM (created on first use).

        Raises ``ClaneHipError`` when no GPU / HIP library is available: there is no CPU path.
        """
        from .engine import SweepEngine
        if self._engine is None:
            dev = _hip.require_gpu(device)
            pg = engine_kwargs.pop("process_group", None)
            if pg is None:
                import torch.distributed as dist
                if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                    pg = dist.group.WORLD
            with torch.cuda.device(dev):
                self._engine = SweepEngine(self.csr, self.X, dev, cosine_mode=cosine_mode, process_group=pg,
                                           **engine_kwargs)
        eng = self._engine
        if self._dirty:
            eng.set_Z(self._Z_host)
            self._dirty = False
        if eng.cosine_mode != cosine_mode:
            eng.cosine_mode, eng.P_valid = cosine_mode, False
        return eng

    def _attach_engine(self, engine) -> None:
        """Use an already-built engine (tests inject one with substitute kernels)."""
        self._engine = engine
        if self._dirty:
            engine.set_Z(self._Z_host)
            self._dirty = False

    # ---- embeddings (graph.py:130-138) ----------------------------------------------------
    def _host_Z(self) -> torch.Tensor:
        if self._engine is not None and not self._dirty:
            return self._engine.get_Z()
        if self._Z_host is None:
            self._Z_host = self.X.clone()      # Vertex.z starts as x (graph.py:19)
        return self._Z_host

    @property
    def Z(self) -> torch.Tensor:
        """Fresh [V, d] tensor; callers may mutate it freely (like the reference's torch.stack)."""
        z = self._host_Z()
        return z if (self._engine is not None and not self._dirty) else z.clone()

    def set_Z(self, Z: torch.Tensor) -> None:
        if tuple(Z.shape) != tuple(self.X.shape):
            raise ValueError(f"set_Z: expected {tuple(self.X.shape)}, got {tuple(Z.shape)}")
        self._Z_host = Z.detach().to("cpu", self.X.dtype).clone()
        self._dirty = True

    def _set_row(self, idx: int, value: torch.Tensor) -> None:
        z = self._host_Z().clone() if not self._dirty else self._Z_host
        z[idx] = value.detach().to("cpu", z.dtype)
        self._Z_host, self._dirty = z, True

    # ---- P (graph.py:118-128) -------------------------------------------------------------
    def build_P(self, similarity) -> torch.Tensor:
        """Row-softmax of per-edge similarity as a coalesced sparse [V, V] tensor (CPU).

        ``CosineSimilarity`` takes the fused HIP path (K0 + K1 + K2, nothing materialised).
        Any other callable follows the plugin protocol literally: it is called ONCE with the
        gathered ``Z[src]``, ``Z[dst]`` batches (GPU tensors) and its scores are normalised by
        the HIP segmented softmax.
        """
        from .similarity import CosineSimilarity
        if isinstance(similarity, CosineSimilarity):
            eng = self.engine(cosine_mode=similarity.mode)
            eng.build_P()
        else:
            # Several GPUs: a plug-in needs WHOLE rows (all d columns of z_src and z_dst), so the rows are divided
            # (halo tables: every row a rank's edges read is in its table) and each rank scores the edges of its own
            # rows.  That is only the reference's single batched call (graph.py:120-121) when a pair's score does not
            # depend on the other pairs of the batch -- the callable says so with `batchwise = True`; a batch-global
            # measure would see 1/N of its batch on every rank and is refused.
            eng = self.engine(exchange=self.PLUGIN_EXCHANGE) if self._engine is None else self.engine()
            if eng.world > 1 and eng.columns:
                raise NotImplementedError(
                    f"custom similarity callables need whole rows of Z; this engine divides the COLUMNS over the GPUs "
                    f"(exchange={eng.exchange!r}). Build the graph's engine with a row division first: "
                    f"graph.engine(exchange='halo') (or 'allgather' / 'allgather_all'), CLI --exchange halo")
            if eng.world > 1 and not getattr(similarity, "batchwise", False):
                raise NotImplementedError(
                    "on several GPUs every rank scores the edges of its own rows, so a custom similarity callable must "
                    "score each pair independently of the rest of the batch and say so with `batchwise = True`; a "
                    "batch-global measure (like the reference's CosineSimilarity, similarity.py:37) is supported on a "
                    "single GPU only")
            rows = torch.repeat_interleave(torch.arange(eng.part.n_local, device=eng.device),
                                           eng.rowptr[1:] - eng.rowptr[:-1])
            Zd = eng.Zcur[:, :eng.d]
            # table row of every own row: a halo table starts with the own rows; a RowPartition places them
            own_pos = (torch.arange(eng.part.n_local, device=eng.device) if eng.halo
                       else torch.from_numpy(eng.part.local_positions()).to(eng.device))
            src_pos = own_pos[rows]
            dst_pos = eng.colidx[:eng.E_loc].long()
            pair_bytes = 2 * eng.E_loc * eng.d * Zd.element_size()          # the two gathered [E, d] batches
            if getattr(similarity, "batchwise", False):
                # the callable vouches that a pair's score does not depend on the other pairs of the batch:
                # gather and score PLUGIN_CHUNK_BYTES at a time (2 x 41 GB at config 3 never exists at once)
                step = max(1, self.PLUGIN_CHUNK_BYTES // max(1, 2 * eng.d * Zd.element_size()))
                for a in range(0, eng.E_loc, step):
                    b = min(a + step, eng.E_loc)
                    part = similarity(Zd[src_pos[a:b]], Zd[dst_pos[a:b]])
                    eng.P[a:b].copy_(part.detach().to(eng.acc_dtype).reshape(-1))
            else:
                limit = self.PLUGIN_SINGLE_CALL_MAX_BYTES
                if limit is None and eng.device.type == "cuda":
                    limit = int(0.8 * torch.cuda.mem_get_info(eng.device)[0])
                if limit is not None and pair_bytes > limit:
                    raise ValueError(
                        f"similarity plugin: the single batched call of the reference's protocol (graph.py:120-121) "
                        f"needs Z[src] and Z[dst] of all {eng.E_loc} edges at once = {pair_bytes / 2**30:.1f} GiB, more "
                        f"than the {limit / 2**30:.1f} GiB available. If a pair's score does not depend on the rest of "
                        f"the batch, set `batchwise = True` on the callable and it is scored in "
                        f"{self.PLUGIN_CHUNK_BYTES >> 20} MiB chunks; a batch-global measure (like the reference's "
                        f"CosineSimilarity) has to come as one call.")
                scores = similarity(Zd[src_pos], Zd[dst_pos]).detach().to(eng.acc_dtype).reshape(-1)
                eng.P[:eng.E_loc].copy_(scores)
            for i, b in enumerate(eng.blocks):        # rows above the engine's threshold: a workgroup per row
                lr = eng.long_rows[i]
                eng.k.segment_softmax(eng.rowptr[b.local_start:], b.nrows, eng.P, 0,
                                      eng.score_threshold if lr is not None else 0, lr)
            eng.P_valid = True
        values = self._gather_P(eng)
        return torch.sparse_coo_tensor(self._edge_index(), values, size=(len(self), len(self)), is_coalesced=True)

    def _gather_P(self, eng) -> torch.Tensor:
        """P values of every rank, put back into the global (row, col)-sorted edge order."""
        out = torch.empty(self.csr.num_edges, dtype=eng.acc_dtype)
        local = (torch.from_numpy(eng.local.edge_origin), eng.P[:eng.E_loc].to("cpu"))
        # rows divided: the ranks that hold the same columns have the rows of P between them
        pieces = [local] if eng.row_world == 1 else eng.comm.all_gather_object(local)
        for origin, vals in pieces:
            out[origin] = vals
        return out
