"""ctypes binding of ``libclane_hip.so`` (C ABI: ``include/clane_hip.h``).

This is the only door between the Python host code and the gfx950 kernels.  There is no
CPU implementation behind it: if the library is missing or a call fails, it raises.

``HipKernels`` exposes one method per ABI entry point, taking torch tensors (device memory
is owned by torch -- plumbing only) and picking the ``_f32`` / ``_f64`` / ``_bf16`` symbol
from the tensor dtype.  Work is enqueued on torch's current HIP stream.
"""
from __future__ import annotations

import abc
import ctypes as C
import os
import threading
from pathlib import Path
from typing import Optional

import torch

LIB_NAME = "libclane_hip.so"
LIB_PATH = Path(__file__).resolve().parent / LIB_NAME
ABI_VERSION = 4

SCORE_REFERENCE, SCORE_PER_EDGE, SCORE_RAW_DOT = 0, 1, 2
SPMM_SINKS_UNTOUCHED = 1
SPMM_TABLE_BEYOND_CACHE = 2
SCORE_FUSE_SOFTMAX = 1
SCORE_MODES = {"reference": SCORE_REFERENCE, "per_edge": SCORE_PER_EDGE, "raw_dot": SCORE_RAW_DOT}

_p, _i64, _i32 = C.c_void_p, C.c_int64, C.c_int32

# symbol -> (restype, argtypes); every symbol include/clane_hip.h declares.
SIGNATURES = {
    "clane_abi_version": (C.c_int, []),
    "clane_last_error": (C.c_char_p, []),
    "clane_build_info": (C.c_char_p, []),
    "clane_xcc_ids": (C.c_int, [_p, _i64, _i32, _p]),
    "clane_check_csr": (C.c_int, [_p, _p, _i64, _i64, _i64, _p, _p]),
    "clane_spmm_partials_len": (_i64, [_i64, _i64]),
    "clane_reduce_ws_len": (_i64, []),
    "clane_reduce_partials": (C.c_int, [_p, _i64, _p, _p, _p]),
    "clane_spmm_split_slab_len": (_i64, [_i64, _i32]),
    "clane_spmm_class_slab_len": (_i64, [_i64, _i32]),
    "clane_device_alloc": (C.c_int, [_i64, C.POINTER(_p)]),
    "clane_device_alloc_contiguous": (C.c_int, [_i64, C.POINTER(_p)]),
    "clane_device_free": (C.c_int, [_p]),
    "clane_ipc_export": (C.c_int, [_p, _p]),
    "clane_ipc_open": (C.c_int, [_p, C.POINTER(_p)]),
    "clane_ipc_close": (C.c_int, [_p]),
}
IPC_HANDLE_BYTES = 64
for _s in ("f32", "f64", "bf16"):
    SIGNATURES[f"clane_row_sqnorm_{_s}"] = (C.c_int, [_p, _i64, _i32, _i64, _p, _p])
    SIGNATURES[f"clane_edge_score_{_s}"] = (
        C.c_int, [_p, _p, _i64, _i64, _p, _i64, _i32, _i32, _p, _p, _p, _i32, _i64, _p, _i64, _p])
    _g = C.c_double if _s == "f64" else C.c_float
    SIGNATURES[f"clane_spmm_update_{_s}"] = (
        C.c_int, [_p, _p, _p, _i64, _i64, _p, _i64, _p, _i64, _g, _p, _i64, _i32, _i64, _i32, _p, _p, _p])
    SIGNATURES[f"clane_spmm_update_long_{_s}"] = (
        C.c_int, [_p, _p, _p, _p, _i64, _i32, _i64, _p, _i64, _p, _i64, _g, _p, _i64, _i32, _p, _p, _p])
    SIGNATURES[f"clane_spmm_update_split_{_s}"] = (
        C.c_int,
        [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _p, _i64, _p, _i64, _g, _p, _i64, _i32, _p, _p, _p, _p])
    SIGNATURES[f"clane_spmm_update_class_{_s}"] = (
        C.c_int,
        [_p, _p, _p, _p, _p, _i64, _i32, _p, _p, _i64, _i64, _p, _i64, _p, _i64, _g, _p, _i64, _i32, _i32, _p, _p, _p, _p])
    SIGNATURES[f"clane_edge_score_class_{_s}"] = (
        C.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i32, _p, _p, _i64, _i64, _p, _i64, _i32, _i32, _p, _p, _p, _i32, _p, _p])
    SIGNATURES[f"clane_gather_rows_{_s}"] = (C.c_int, [_p, _i64, _p, _i64, _i32, _p, _i64, _p])
    SIGNATURES[f"clane_l1_distance_{_s}"] = (C.c_int, [_p, _i64, _p, _i64, _i64, _i32, _p, _p, _p, _p])
for _s in ("f32", "f64"):
    SIGNATURES[f"clane_degree_weighted_sums_{_s}"] = (C.c_int, [_p, _p, _p, _i64, _p, _p, _p])
    SIGNATURES[f"clane_edge_score_finalize_{_s}"] = (C.c_int, [_p, _p, _i64, _i64, _i32, _p, _p, _p, _p])
    SIGNATURES[f"clane_segment_softmax_{_s}"] = (C.c_int, [_p, _i64, _p, _i64, _i64, _p, _i64, _p])
    SIGNATURES[f"clane_pair_cosine_{_s}"] = (C.c_int, [_p, _i64, _p, _i64, _i64, _i32, _p, _p, _p])

_SUFFIX = {torch.float32: "f32", torch.float64: "f64", torch.bfloat16: "bf16"}
_ACC = {torch.float32: torch.float32, torch.float64: torch.float64, torch.bfloat16: torch.float32}
VEC_ELEMS = {torch.float32: 4, torch.float64: 2, torch.bfloat16: 8}  # elements per 16-byte pack


class ClaneHipError(RuntimeError):
    pass


class _MirrorStruct(C.Structure):       # clane_mirror_t
    _fields_ = [("row_ptr", _p), ("slot", _p), ("bufs", _p), ("ld", _i64), ("aligned16", _i32)]


MIRROR_ROW_BITS = 28


class Mirror:
    """Further destinations of the rows an spmm_update* call finishes (``clane_mirror_t``): row r of the call is
    also stored at the places ``slot[row_ptr[r]:row_ptr[r+1]]``, a place being ``buffer << 28 | row`` into
    ``bufs`` (one tensor, or up to 8 of equal leading dimension -- e.g. other GPUs' tables).  Holds them alive."""

    def __init__(self, row_ptr: torch.Tensor, slot: torch.Tensor, bufs):
        if row_ptr.dtype != torch.int64 or slot.dtype != torch.int32 or not (row_ptr.is_contiguous()
                                                                             and slot.is_contiguous()):
            raise ValueError("Mirror: row_ptr must be contiguous int64 and slot contiguous int32")
        bufs = [bufs] if isinstance(bufs, (torch.Tensor, PeerMatrix)) else list(bufs)
        if not 1 <= len(bufs) <= 8:
            raise ValueError("Mirror: 1 to 8 destination buffers")
        mats = [_mat(b, "mirror buffer") for b in bufs]
        if len({ld for _, ld in mats}) != 1 or len({b.dtype for b in bufs}) != 1:
            raise ValueError("Mirror: the buffers must share dtype and leading dimension")
        if max(b.shape[0] for b in bufs) > (1 << MIRROR_ROW_BITS):
            raise ValueError("Mirror: a buffer has more than 2^28 rows")
        self.row_ptr, self.slot, self.bufs = row_ptr, slot, bufs
        self.buf = bufs[0]
        self.bases = torch.tensor([ptr for ptr, _ in mats], dtype=torch.int64, device=row_ptr.device)
        self.c = _MirrorStruct(row_ptr.data_ptr(), slot.data_ptr(), self.bases.data_ptr(), mats[0][1],
                               int(all(ptr % 16 == 0 for ptr, _ in mats)))


class _RawDeviceMemory:
    """What torch.as_tensor needs to view foreign device memory without copying it."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class PeerMatrix:
    """Address + shape of a row-major matrix in ANOTHER process's (GPU's) memory, mapped here.  Not a torch tensor
    on purpose: torch would attribute it to the owning device; only kernels of this library touch it, by address."""

    def __init__(self, ptr: int, shape, dtype: torch.dtype):
        self._ptr, self.shape, self.dtype = ptr, tuple(shape), dtype

    def data_ptr(self) -> int:
        return self._ptr

    def dim(self) -> int:
        return len(self.shape)

    def stride(self, i: Optional[int] = None):
        st = (self.shape[1], 1)
        return st if i is None else st[i]


class DeviceBuffer:
    """A matrix in device memory that is its OWN allocation (hipMalloc through the C ABI), so that another process
    can map it (hipIpc works on allocation bases; torch sub-allocates) -- or such a mapping of another process's
    matrix (`DeviceBuffer.open`).  `.tensor` views it; torch neither owns nor frees it: this object does."""

    def __init__(self, lib, shape, dtype: torch.dtype, device, _mapped_from: Optional[bytes] = None,
                 contiguous: bool = False):
        """``contiguous``: physically contiguous backing (clane_device_alloc_contiguous); raises ClaneHipError when
        the driver has no such range free -- the caller falls back to an ordinary allocation."""
        self.lib, self.shape, self.dtype = lib, tuple(int(x) for x in shape), dtype
        self.nbytes = max(16, int(torch.empty(0, dtype=dtype).element_size()) * int(torch.Size(self.shape).numel()))
        self.mapped = _mapped_from is not None
        ptr = _p()
        with torch.cuda.device(device):
            if self.mapped:
                rc = lib.clane_ipc_open(_mapped_from, C.byref(ptr))
            elif contiguous:
                rc = lib.clane_device_alloc_contiguous(self.nbytes, C.byref(ptr))
            else:
                rc = lib.clane_device_alloc(self.nbytes, C.byref(ptr))
        if rc != 0:
            raise ClaneHipError(f"{'clane_ipc_open' if self.mapped else 'clane_device_alloc'} failed ({rc}): "
                                f"{lib.clane_last_error().decode()}")
        self.ptr = ptr.value
        if self.mapped:
            self.tensor = PeerMatrix(self.ptr, self.shape, dtype)
        else:
            raw = torch.as_tensor(_RawDeviceMemory(self.ptr, self.nbytes), device=torch.device(device))
            n = int(torch.Size(self.shape).numel()) * torch.empty(0, dtype=dtype).element_size()
            self.tensor = raw[:n].view(dtype).view(self.shape)
            self.tensor.zero_()

    def export(self) -> bytes:
        handle = C.create_string_buffer(IPC_HANDLE_BYTES)
        rc = self.lib.clane_ipc_export(self.ptr, handle)
        if rc != 0:
            raise ClaneHipError(f"clane_ipc_export failed ({rc}): {self.lib.clane_last_error().decode()}")
        return handle.raw

    @classmethod
    def open(cls, lib, handle: bytes, shape, dtype: torch.dtype, device) -> "DeviceBuffer":
        return cls(lib, shape, dtype, device, _mapped_from=handle)

    def close(self) -> None:
        if self.ptr:
            (self.lib.clane_ipc_close if self.mapped else self.lib.clane_device_free)(self.ptr)
            self.ptr = 0

    def __del__(self):
        try:
            self.close()
        except Exception:       # interpreter shutdown: the process is about to give everything back anyway
            pass


def _mirror_arg(mirror: Optional["Mirror"], dtype: torch.dtype):
    if mirror is None:
        return None
    if mirror.buf.dtype != dtype:
        raise ValueError("mirror.buf must have the dtype of Z_new")
    return C.cast(C.pointer(mirror.c), _p)


def load_library(path: Optional[Path] = None) -> C.CDLL:
    """dlopen the library and bind every declared symbol.  Raises if anything is missing."""
    if path is None and os.environ.get("CLANE_HIP_LIB"):      # A/B builds of the same ABI (tools/ab_variants.py)
        path = os.environ["CLANE_HIP_LIB"]
    path = Path(path) if path is not None else LIB_PATH
    if not path.exists():
        raise ClaneHipError(
            f"{path} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (or `make -C clane_amd/csrc`) -- clane_amd has no CPU fallback.")
    lib = C.CDLL(str(path))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype, fn.argtypes = res, args
    got = lib.clane_abi_version()
    if got != ABI_VERSION:
        raise ClaneHipError(f"{path}: ABI version {got}, expected {ABI_VERSION}")
    return lib


def acc_dtype(dtype: torch.dtype) -> torch.dtype:
    """dtype of P / scores / squared norms for a given storage dtype of Z."""
    try:
        return _ACC[dtype]
    except KeyError:
        raise TypeError(f"clane_amd supports float32, float64 and bfloat16 embeddings, not {dtype}") from None


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _mat(t: torch.Tensor, name: str):
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"{name}: need a row-major 2-D tensor with unit column stride, got {tuple(t.shape)} "
                         f"strides {t.stride()}")
    return t.data_ptr(), t.stride(0)


def _vec(t: torch.Tensor, dtype: torch.dtype, name: str):
    if t.dtype != dtype or not t.is_contiguous():
        raise ValueError(f"{name}: need a contiguous {dtype} tensor, got {t.dtype} contiguous={t.is_contiguous()}")
    return t.data_ptr()


class KernelBackend(abc.ABC):
    """Every call ``SweepEngine`` / ``Graph`` / ``CosineSimilarity`` make on their kernel object -- the contract
    between the host logic and whatever executes the kernels.  ``HipKernels`` (the C ABI) is the product's only
    implementation; the CPU suite's test double (tests/oracle_kernels.py) implements the same list.  An
    implementation that lacks one of these cannot be instantiated, so nothing the engine relies on (the CSR check
    that keeps a bad index from faulting the GPU, say) can be skipped silently: the engine calls them
    unconditionally."""

    # sizes
    @abc.abstractmethod
    def spmm_partials_len(self, nrows: int, n_long: int) -> int: ...
    @abc.abstractmethod
    def reduce_ws_len(self) -> int: ...
    @abc.abstractmethod
    def spmm_split_slab_len(self, n_segments: int, d: int) -> int: ...
    @abc.abstractmethod
    def spmm_class_slab_len(self, n_slots: int, d: int) -> int: ...
    # input check, build description
    @abc.abstractmethod
    def check_csr(self, rowptr, colidx, nrows: int, n_edges: int, table_rows: int) -> None: ...
    @abc.abstractmethod
    def build_info(self) -> str: ...
    # K0 / K1 / K2
    @abc.abstractmethod
    def row_sqnorm(self, Z, d: int, sq): ...
    @abc.abstractmethod
    def degree_weighted_sums(self, sq, rowptr, indeg, nrows: int, ws, out2): ...
    @abc.abstractmethod
    def edge_score(self, rowptr, colidx, nrows, row0, Z, d, mode, sums2, sq, scores, long_threshold=0, long_rows=None,
                   fuse_softmax=False): ...
    @abc.abstractmethod
    def edge_score_class(self, rowptr, colidx, item_e0, item_len, item_slot, item_row, items_per_block, class_rows,
                         slot_ptr, row0, Z, d, mode, sums2, sq, scores, stats=None, fuse_softmax=False,
                         n_slots=None, row_parts=1): ...
    @abc.abstractmethod
    def edge_score_finalize(self, rowptr, colidx, nrows, row0, mode, sums2, sq, scores): ...
    @abc.abstractmethod
    def segment_softmax(self, rowptr, nrows, vals, min_degree=0, max_degree=0, long_rows=None): ...
    @abc.abstractmethod
    def pair_cosine(self, A, B, d, out, ws): ...
    # K3
    @abc.abstractmethod
    def spmm_update(self, rowptr, colidx, P, nrows, row0, Z_old, X, gamma, Z_new, d, long_threshold, partials,
                    sinks_untouched=False, mirror=None, beyond_cache=False): ...
    @abc.abstractmethod
    def spmm_update_long(self, rowptr, colidx, P, long_rows, waves_per_row, row0, Z_old, X, gamma, Z_new, d, partials,
                         mirror=None): ...
    @abc.abstractmethod
    def spmm_update_split(self, rowptr, colidx, P, split_rows, seg_ptr, seg_row, edges_per_segment, row0, Z_old, X,
                          gamma, Z_new, d, slab, partials, mirror=None): ...
    @abc.abstractmethod
    def spmm_update_class(self, colidx, P, item_e0, item_len, item_slot, items_per_block, class_rows, slot_ptr, row0,
                          Z_old, X, gamma, Z_new, d, slab, partials, mirror=None, beyond_cache=False): ...
    @abc.abstractmethod
    def reduce_partials(self, partials, n, ws, out): ...
    @abc.abstractmethod
    def l1_distance(self, A, B, d, ws, out, sq_a=None): ...
    @abc.abstractmethod
    def gather_rows(self, src, idx, d, dst): ...
    # further destinations of finished rows, tables other processes map
    @abc.abstractmethod
    def make_mirror(self, row_ptr, slot, bufs): ...
    @abc.abstractmethod
    def shareable_matrix(self, shape, dtype, device): ...
    @abc.abstractmethod
    def contiguous_matrix(self, shape, dtype, device): ...
    @abc.abstractmethod
    def open_shared_matrix(self, handle, shape, dtype, device): ...

    def bind(self, method: str, *args, **kwargs):
        """A zero-argument callable that makes the call ``method(*args, **kwargs)``; an implementation may
        pre-marshal it (HipKernels does)."""
        fn = getattr(self, method)
        return lambda: fn(*args, **kwargs)


class HipKernels(KernelBackend):
    """Tensor-level view of the C ABI.  One instance per process is enough (stateless)."""

    def __init__(self, lib: Optional[C.CDLL] = None):
        self.lib = lib if lib is not None else load_library()
        self._tls = threading.local()          # bind() records per thread: one instance serves every thread

    # -- helpers ------------------------------------------------------------------------
    def _invoke(self, fn, what: str, *cargs):
        """Call an ABI function now -- or, inside bind(), only record the bound call."""
        rec = getattr(self._tls, "recording", None)
        if rec is not None:
            rec.append((fn, cargs, what))
        else:
            self._check(fn(*cargs), what)

    def bind(self, method: str, *args, **kwargs):
        """The ABI call `method(*args, **kwargs)` would make, pre-marshalled: returns a zero-argument callable.
        Pointers, sizes and the CURRENT stream are captured now, so a sweep can replay a flat list of such
        calls without re-slicing tensors or re-checking arguments (host time per launch: ~3 us instead of ~25)."""
        self._tls.recording = []
        try:
            getattr(self, method)(*args, **kwargs)
            (fn, cargs, what), = self._tls.recording
        finally:
            self._tls.recording = None
        check = self._check
        return lambda: check(fn(*cargs), what)

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise ClaneHipError(f"{what} failed ({rc}): {self.lib.clane_last_error().decode()}")

    @staticmethod
    def _stream(t: torch.Tensor):
        if not t.is_cuda:
            raise ClaneHipError("HipKernels needs tensors in GPU memory (got a CPU tensor); there is no CPU path")
        return torch.cuda.current_stream(t.device).cuda_stream

    def _fn(self, base: str, dtype: torch.dtype):
        try:
            return getattr(self.lib, f"{base}_{_SUFFIX[dtype]}")
        except KeyError:
            raise TypeError(f"{base}: unsupported dtype {dtype}") from None

    # -- sizes --------------------------------------------------------------------------
    def spmm_partials_len(self, nrows: int, n_long: int) -> int:
        return int(self.lib.clane_spmm_partials_len(nrows, n_long))

    def xcc_ids(self, n_blocks: int, block_threads: int = 256, device=None) -> torch.Tensor:
        """XCD each workgroup of an n_blocks-workgroup launch ran on (int32 tensor on the device)."""
        out = torch.full((n_blocks,), -1, dtype=torch.int32, device=device if device is not None else "cuda")
        self._check(self.lib.clane_xcc_ids(out.data_ptr(), n_blocks, block_threads, self._stream(out)), "clane_xcc_ids")
        return out

    def check_csr(self, rowptr: torch.Tensor, colidx: torch.Tensor, nrows: int, n_edges: int, table_rows: int) -> None:
        """Raise ValueError unless rowptr[0..nrows] is non-decreasing within [0, n_edges] and colidx[:n_edges] holds
        rows of a table of ``table_rows`` rows (one pass on the device; blocks until it is done)."""
        if rowptr.dtype != torch.int64 or colidx.dtype != torch.int32 or rowptr.numel() < nrows + 1 or colidx.numel() < n_edges:
            raise ValueError("check_csr: rowptr must be int64 [nrows + 1], colidx int32 [n_edges]")
        status = torch.zeros(1, dtype=torch.int32, device=rowptr.device)
        self._check(self.lib.clane_check_csr(rowptr.data_ptr(), colidx.data_ptr(), nrows, n_edges, table_rows,
                                             status.data_ptr(), self._stream(rowptr)), "clane_check_csr")
        bad = int(status.item())
        if bad:
            what = [w for bit, w in ((1, "rowptr is not a non-decreasing sequence within [0, n_edges]"),
                                     (2, f"colidx holds entries outside [0, {table_rows})")) if bad & bit]
            raise ValueError("the CSR handed to the kernels is not valid: " + "; ".join(what))

    def build_info(self) -> str:
        return self.lib.clane_build_info().decode()

    def make_mirror(self, row_ptr, slot, bufs) -> Mirror:
        return Mirror(row_ptr, slot, bufs)

    def reduce_ws_len(self) -> int:
        return int(self.lib.clane_reduce_ws_len())

    def contiguous_matrix(self, shape, dtype: torch.dtype, device) -> DeviceBuffer:
        """A zeroed matrix of its own, physically contiguous allocation (raises ClaneHipError when none is to be had)."""
        return DeviceBuffer(self.lib, shape, dtype, device, contiguous=True)

    def shareable_matrix(self, shape, dtype: torch.dtype, device) -> DeviceBuffer:
        """Zero-filled device matrix that other processes can map (DeviceBuffer.export / .open)."""
        return DeviceBuffer(self.lib, shape, dtype, device)

    def open_shared_matrix(self, handle: bytes, shape, dtype: torch.dtype, device) -> DeviceBuffer:
        return DeviceBuffer.open(self.lib, handle, shape, dtype, device)

    # -- K0 -----------------------------------------------------------------------------
    def row_sqnorm(self, Z: torch.Tensor, d: int, sq: torch.Tensor):
        zp, ldz = _mat(Z, "Z")
        self._check(self._fn("clane_row_sqnorm", Z.dtype)(
            zp, Z.shape[0], d, ldz, _vec(sq, acc_dtype(Z.dtype), "sq"), self._stream(Z)), "clane_row_sqnorm")

    def degree_weighted_sums(self, sq, rowptr, indeg, nrows: int, ws, out2):
        self._check(self._fn("clane_degree_weighted_sums", sq.dtype)(
            _ptr(sq), _vec(rowptr, torch.int64, "rowptr"), _vec(indeg, torch.int32, "indeg"), nrows,
            _vec(ws, torch.float64, "ws"), _vec(out2, torch.float64, "out2"), self._stream(out2)),
            "clane_degree_weighted_sums")

    # -- K1 / K2 ------------------------------------------------------------------------
    def edge_score(self, rowptr, colidx, nrows: int, row0: int, Z, d: int, mode: int, sums2, sq, scores,
                   long_threshold: int = 0, long_rows=None, fuse_softmax: bool = False):
        zp, ldz = _mat(Z, "Z")
        n_long = 0 if long_rows is None else long_rows.numel()
        self._check(self._fn("clane_edge_score", Z.dtype)(
            _vec(rowptr, torch.int64, "rowptr"), _vec(colidx, torch.int32, "colidx"), nrows, row0, zp, ldz, d, mode,
            _ptr(sums2), _ptr(sq), _vec(scores, acc_dtype(Z.dtype), "scores"),
            SCORE_FUSE_SOFTMAX if fuse_softmax else 0, long_threshold,
            None if n_long == 0 else _vec(long_rows, torch.int32, "long_rows"), n_long,
            self._stream(Z)), "clane_edge_score")

    def edge_score_class(self, rowptr, colidx, item_e0, item_len, item_slot, item_row, items_per_block: int, class_rows,
                         slot_ptr, row0: int, Z, d: int, mode: int, sums2, sq, scores, stats=None,
                         fuse_softmax: bool = False, n_slots: Optional[int] = None, row_parts: int = 1):
        """K1 over the class rows' work items (XCD-affine gathers); with `fuse_softmax` every listed row leaves
        soft-maxed (stats: 2 accumulate-type elements per slot).  ``n_slots`` = slot_ptr[-1] when the caller knows it
        on the host (the engine does): without it the size check of `stats` reads it back from the device, which
        blocks the host on everything queued before this call."""
        zp, ldz = _mat(Z, "Z")
        n_items = item_e0.numel()
        if n_items % items_per_block or any(t.numel() != n_items for t in (item_len, item_slot, item_row)):
            raise ValueError("edge_score_class: the item arrays must hold whole blocks of items_per_block items")
        if fuse_softmax and (stats is None or stats.numel() < 2 * (int(slot_ptr[-1]) if n_slots is None else n_slots)):
            raise ValueError("edge_score_class: stats needs 2 elements per slot")
        acc = acc_dtype(Z.dtype)
        self._check(self._fn("clane_edge_score_class", Z.dtype)(
            _vec(rowptr, torch.int64, "rowptr"), _vec(colidx, torch.int32, "colidx"),
            _vec(item_e0, torch.int64, "item_e0"), _vec(item_len, torch.int32, "item_len"),
            _vec(item_slot, torch.int32, "item_slot"), _vec(item_row, torch.int32, "item_row"),
            n_items // items_per_block, items_per_block, _vec(class_rows, torch.int32, "class_rows"),
            _vec(slot_ptr, torch.int64, "slot_ptr"), class_rows.numel(), row0, zp, ldz, d, mode, _ptr(sums2), _ptr(sq),
            _vec(scores, acc, "scores"),
            (SCORE_FUSE_SOFTMAX | ((max(1, min(255, int(row_parts))) & 0xff) << 8)) if fuse_softmax else 0,
            None if stats is None else _vec(stats, acc, "stats"), self._stream(Z)), "clane_edge_score_class")

    def edge_score_finalize(self, rowptr, colidx, nrows: int, row0: int, mode: int, sums2, sq, scores):
        """RAW_DOT scores summed over the GPUs of a column-split run -> scores of `mode`, in place."""
        self._check(self._fn("clane_edge_score_finalize", scores.dtype)(
            _vec(rowptr, torch.int64, "rowptr"), _vec(colidx, torch.int32, "colidx"), nrows, row0, mode,
            _ptr(sums2), _ptr(sq), scores.data_ptr(), self._stream(scores)), "clane_edge_score_finalize")

    def segment_softmax(self, rowptr, nrows: int, vals, min_degree: int = 0, max_degree: int = 0, long_rows=None):
        n_long = 0 if long_rows is None else long_rows.numel()
        self._check(self._fn("clane_segment_softmax", vals.dtype)(
            _vec(rowptr, torch.int64, "rowptr"), nrows, vals.data_ptr(), min_degree, max_degree,
            None if n_long == 0 else _vec(long_rows, torch.int32, "long_rows"), n_long, self._stream(vals)),
            "clane_segment_softmax")

    # -- K3 -----------------------------------------------------------------------------
    @staticmethod
    def _sq_arg(sq: Optional[torch.Tensor], dtype: torch.dtype):
        return None if sq is None else _vec(sq, acc_dtype(dtype), "sq_a")

    def spmm_update(self, rowptr, colidx, P, nrows: int, row0: int, Z_old, X, gamma: float, Z_new, d: int,
                    long_threshold: int, partials, sinks_untouched: bool = False, mirror: Optional[Mirror] = None,
                    beyond_cache: bool = False):
        """Main pass: every row of <= long_threshold edges (0 = all rows).  With `sinks_untouched` rows
        without out-edges are neither read nor written (caller keeps Z_new == Z_old there).  `beyond_cache`: the hint
        CLANE_SPMM_TABLE_BEYOND_CACHE (the table is far beyond the caches; results do not depend on it)."""
        zo, ldz = _mat(Z_old, "Z_old")
        xp, ldx = _mat(X, "X")
        zn, ldo = _mat(Z_new, "Z_new")
        if not (Z_old.dtype == X.dtype == Z_new.dtype):
            raise ValueError("spmm_update: Z_old, X, Z_new must share a dtype")
        self._invoke(self._fn("clane_spmm_update", Z_old.dtype), "clane_spmm_update",
                     _vec(rowptr, torch.int64, "rowptr"), _vec(colidx, torch.int32, "colidx"),
                     _vec(P, acc_dtype(Z_old.dtype), "P"), nrows, row0, zo, ldz, xp, ldx, gamma, zn, ldo, d,
                     long_threshold, (SPMM_SINKS_UNTOUCHED if sinks_untouched else 0) | (SPMM_TABLE_BEYOND_CACHE if beyond_cache else 0),
                     _mirror_arg(mirror, Z_new.dtype), _vec(partials, torch.float64, "partials"), self._stream(Z_old))

    def spmm_update_long(self, rowptr, colidx, P, long_rows, waves_per_row: int, row0: int, Z_old, X, gamma: float,
                         Z_new, d: int, partials, mirror: Optional[Mirror] = None):
        """Row-split pass: one workgroup of `waves_per_row` (4 | 16) waves per listed row; writes
        long_rows.numel() partials."""
        zo, ldz = _mat(Z_old, "Z_old")
        xp, ldx = _mat(X, "X")
        zn, ldo = _mat(Z_new, "Z_new")
        if not (Z_old.dtype == X.dtype == Z_new.dtype):
            raise ValueError("spmm_update_long: Z_old, X, Z_new must share a dtype")
        self._invoke(self._fn("clane_spmm_update_long", Z_old.dtype), "clane_spmm_update_long",
                     _vec(rowptr, torch.int64, "rowptr"), _vec(colidx, torch.int32, "colidx"),
                     _vec(P, acc_dtype(Z_old.dtype), "P"), _vec(long_rows, torch.int32, "long_rows"),
                     long_rows.numel(), waves_per_row, row0, zo, ldz, xp, ldx, gamma, zn, ldo, d,
                     _mirror_arg(mirror, Z_new.dtype), _vec(partials, torch.float64, "partials"), self._stream(Z_old))

    def spmm_split_slab_len(self, n_segments: int, d: int) -> int:
        return int(self.lib.clane_spmm_split_slab_len(n_segments, d))

    def spmm_update_split(self, rowptr, colidx, P, split_rows, seg_ptr, seg_row, edges_per_segment: int, row0: int,
                          Z_old, X, gamma: float, Z_new, d: int, slab, partials, mirror: Optional[Mirror] = None):
        """Hub rows cut into segments over several workgroups + fixed-order combine; writes split_rows.numel()
        partials."""
        zo, ldz = _mat(Z_old, "Z_old")
        xp, ldx = _mat(X, "X")
        zn, ldo = _mat(Z_new, "Z_new")
        self._invoke(self._fn("clane_spmm_update_split", Z_old.dtype), "clane_spmm_update_split",
                     _vec(rowptr, torch.int64, "rowptr"), _vec(colidx, torch.int32, "colidx"),
                     _vec(P, acc_dtype(Z_old.dtype), "P"), _vec(split_rows, torch.int32, "split_rows"),
                     _vec(seg_ptr, torch.int64, "seg_ptr"), _vec(seg_row, torch.int32, "seg_row"),
                     split_rows.numel(), seg_row.numel(), edges_per_segment, row0, zo, ldz, xp, ldx, gamma, zn, ldo, d,
                     _vec(slab, acc_dtype(Z_old.dtype), "slab"), _mirror_arg(mirror, Z_new.dtype),
                     _vec(partials, torch.float64, "partials"), self._stream(Z_old))

    def spmm_class_slab_len(self, n_slots: int, d: int) -> int:
        return int(self.lib.clane_spmm_class_slab_len(n_slots, d))

    def spmm_update_class(self, colidx, P, item_e0, item_len, item_slot, items_per_block: int, class_rows, slot_ptr,
                          row0: int, Z_old, X, gamma: float, Z_new, d: int, slab, partials,
                          mirror: Optional[Mirror] = None, beyond_cache: bool = False):
        """XCD-affine pass over the listed long rows (edges sorted by (XCD class of the column, column), cut into items; item
        blocks of class b at block index 8 j + b) + fixed-order combine; writes class_rows.numel() partials."""
        zo, ldz = _mat(Z_old, "Z_old")
        xp, ldx = _mat(X, "X")
        zn, ldo = _mat(Z_new, "Z_new")
        n_items = item_e0.numel()
        if n_items % items_per_block or item_len.numel() != n_items or item_slot.numel() != n_items:
            raise ValueError("spmm_update_class: the item arrays must hold whole blocks of items_per_block items")
        self._invoke(self._fn("clane_spmm_update_class", Z_old.dtype), "clane_spmm_update_class",
                     _vec(colidx, torch.int32, "colidx"), _vec(P, acc_dtype(Z_old.dtype), "P"),
                     _vec(item_e0, torch.int64, "item_e0"), _vec(item_len, torch.int32, "item_len"),
                     _vec(item_slot, torch.int32, "item_slot"), n_items // items_per_block, items_per_block,
                     _vec(class_rows, torch.int32, "class_rows"), _vec(slot_ptr, torch.int64, "slot_ptr"),
                     class_rows.numel(), row0, zo, ldz, xp, ldx, gamma, zn, ldo, d,
                     SPMM_TABLE_BEYOND_CACHE if beyond_cache else 0,
                     _vec(slab, acc_dtype(Z_old.dtype), "slab"), _mirror_arg(mirror, Z_new.dtype),
                     _vec(partials, torch.float64, "partials"), self._stream(Z_old))

    def reduce_partials(self, partials, n: int, ws, out):
        self._invoke(self.lib.clane_reduce_partials, "clane_reduce_partials",
                     _vec(partials, torch.float64, "partials"), n, _vec(ws, torch.float64, "ws"),
                     _vec(out, torch.float64, "out"), self._stream(out))

    def l1_distance(self, A, B, d: int, ws, out, sq_a: Optional[torch.Tensor] = None):
        """out[0] = sum|A - B|; with `sq_a` also the squared norm of every row of A (bit for bit row_sqnorm's)."""
        ap, lda = _mat(A, "A")
        bp, ldb = _mat(B, "B")
        self._check(self._fn("clane_l1_distance", A.dtype)(
            ap, lda, bp, ldb, A.shape[0], d, self._sq_arg(sq_a, A.dtype), _vec(ws, torch.float64, "ws"),
            _vec(out, torch.float64, "out"), self._stream(A)), "clane_l1_distance")

    def gather_rows(self, src, idx, d: int, dst):
        """dst[i, :] = src[idx[i], :] (send-buffer packing of the halo exchange)."""
        sp, lds = _mat(src, "src")
        dp, ldd = _mat(dst, "dst")
        self._invoke(self._fn("clane_gather_rows", src.dtype), "clane_gather_rows",
                     sp, lds, _vec(idx, torch.int32, "idx"), idx.numel(), d, dp, ldd, self._stream(src))

    # -- CosineSimilarity on explicit pairs ------------------------------------------------
    def pair_cosine(self, A, B, d: int, out, ws):
        ap, lda = _mat(A, "A")
        bp, ldb = _mat(B, "B")
        self._check(self._fn("clane_pair_cosine", A.dtype)(
            ap, lda, bp, ldb, A.shape[0], d, _vec(out, acc_dtype(A.dtype), "out"), _vec(ws, torch.float64, "ws"),
            self._stream(A)), "clane_pair_cosine")


_kernels: Optional[HipKernels] = None


def kernels() -> HipKernels:
    """Process-wide HipKernels; raises ClaneHipError when the library is not built."""
    global _kernels
    if _kernels is None:
        _kernels = HipKernels()
    return _kernels


def require_gpu(device=None) -> torch.device:
    """The device the hot path runs on.  Raises (never falls back) when there is none."""
    if not torch.cuda.is_available():
        raise ClaneHipError("clane_amd runs its embedding loop in HIP kernels on an MI355X; no GPU is visible "
                            "to this process and there is no CPU fallback.")
    if device is None or torch.device(device).type != "cuda":
        return torch.device("cuda", torch.cuda.current_device())
    dev = torch.device(device)
    return dev if dev.index is not None else torch.device("cuda", torch.cuda.current_device())
