"""TEST DOUBLE for ``clane_amd._hip.HipKernels`` backed by the CPU oracle.

Lives under tests/ on purpose: it lets the CPU suite drive the *host* logic of clane_amd
(SweepEngine launch sequence, row partition, all-gather layout, Embedder control flow, CLI)
without a GPU.  It is never importable from the product package, and the product never
substitutes it: ``SweepEngine`` only gets it when a test passes it in explicitly.
"""
import numpy as np
import torch

from clane_amd._hip import KernelBackend
from clane_amd.xcd import xcd_class
from oracle import clane_oracle as O


def _np(t):
    return t.detach().cpu().numpy()


class OracleKernels(KernelBackend):
    """Implements the whole ``KernelBackend`` contract (the abstract base refuses to instantiate otherwise), so the
    engine never has to ask whether a call exists."""

    def build_info(self):
        return "substitute kernels (tests/oracle_kernels.py)"

    def open_shared_matrix(self, handle, shape, dtype, device):
        raise NotImplementedError("one process: ThreadComm.share_matrices hands the tensors over directly")

    def check_csr(self, rowptr, colidx, nrows, n_edges, table_rows):
        rp, ci = rowptr.cpu().numpy()[:nrows + 1], colidx.cpu().numpy()[:n_edges]
        if rp[0] < 0 or rp[-1] > n_edges or (np.diff(rp) < 0).any():
            raise ValueError("the CSR handed to the kernels is not valid: rowptr")
        if n_edges and (ci.min() < 0 or ci.max() >= table_rows):
            raise ValueError("the CSR handed to the kernels is not valid: colidx")

    def spmm_partials_len(self, nrows, n_long):
        return -(-max(nrows, 1) // 32) + n_long

    def reduce_ws_len(self):
        return 8

    def row_sqnorm(self, Z, d, sq):
        sq.copy_(Z[:, :d].to(sq.dtype).pow(2).sum(1))

    def degree_weighted_sums(self, sq, rowptr, indeg, nrows, ws, out2):
        outdeg = (rowptr[1:nrows + 1] - rowptr[:nrows]).double()
        out2[0] = (outdeg * sq[:nrows].double()).sum()
        out2[1] = (indeg[:nrows].double() * sq[:nrows].double()).sum()

    def edge_score(self, rowptr, colidx, nrows, row0, Z, d, mode, sums2, sq, scores, long_threshold=0,
                   long_rows=None, fuse_softmax=False):
        rp = _np(rowptr[:nrows + 1])
        if rp[-1] == rp[0]:
            return
        rows = torch.from_numpy(np.repeat(np.arange(nrows), np.diff(rp))) + row0
        cols = colidx[rp[0]:rp[-1]].long()
        Zf = Z[:, :d].to(scores.dtype)
        dots = (Zf[rows] * Zf[cols]).sum(1)
        if mode == 0:
            dots = dots / (sums2[0].to(scores.dtype).sqrt() * sums2[1].to(scores.dtype).sqrt())
        elif mode == 1:
            dots = dots / (sq[rows].sqrt() * sq[cols].sqrt())
        scores[rp[0]:rp[-1]] = dots
        if fuse_softmax:      # every row the call scores
            for r in range(nrows):
                if rp[r + 1] > rp[r]:
                    scores[rp[r]:rp[r + 1]] = torch.softmax(scores[rp[r]:rp[r + 1]], 0)

    def edge_score_class(self, rowptr, colidx, item_e0, item_len, item_slot, item_row, items_per_block, class_rows,
                         slot_ptr, row0, Z, d, mode, sums2, sq, scores, stats=None, fuse_softmax=False, n_slots=None,
                         row_parts=1):
        """K1 over the class rows' items; checks the same layout contract as spmm_update_class."""
        assert n_slots is None or n_slots == int(slot_ptr[-1])
        e0, ln, rw = _np(item_e0), _np(item_len), _np(item_row)
        assert 4 <= items_per_block <= 64 and e0.size % items_per_block == 0 and e0.size // items_per_block % 8 == 0
        Zf = Z[:, :d].to(scores.dtype)
        listed = set(class_rows.tolist())
        for k in range(e0.size):
            if ln[k] == 0:
                continue
            a, b = int(e0[k]), int(e0[k]) + int(ln[k])
            cols = colidx[a:b].long()
            assert bool((xcd_class(cols) == (k // items_per_block) % 8).all()) and int(rw[k]) in listed
            src = row0 + int(rw[k])
            dots = (Zf[src].unsqueeze(0) * Zf[cols]).sum(1)
            if mode == 0:
                dots = dots / (sums2[0].to(scores.dtype).sqrt() * sums2[1].to(scores.dtype).sqrt())
            elif mode == 1:
                dots = dots / (sq[src].sqrt() * sq[cols].sqrt())
            scores[a:b] = dots
        if fuse_softmax:
            rp = _np(rowptr)
            for r in class_rows.tolist():
                scores[rp[r]:rp[r + 1]] = torch.softmax(scores[rp[r]:rp[r + 1]], 0)

    def edge_score_finalize(self, rowptr, colidx, nrows, row0, mode, sums2, sq, scores):
        rp = _np(rowptr[:nrows + 1])
        if rp[-1] == rp[0] or mode == 2:
            return
        view = scores[rp[0]:rp[-1]]
        if mode == 0:
            view /= sums2[0].to(scores.dtype).sqrt() * sums2[1].to(scores.dtype).sqrt()
        else:
            rows = torch.from_numpy(np.repeat(np.arange(nrows), np.diff(rp))) + row0
            view /= sq[rows].sqrt() * sq[colidx[rp[0]:rp[-1]].long()].sqrt()

    def segment_softmax(self, rowptr, nrows, vals, min_degree=0, max_degree=0, long_rows=None):
        rp = _np(rowptr[:nrows + 1])
        for r in range(nrows):
            if rp[r + 1] - rp[r] > min_degree:
                vals[rp[r]:rp[r + 1]] = torch.softmax(vals[rp[r]:rp[r + 1]], 0)

    def make_mirror(self, row_ptr, slot, bufs):
        return (row_ptr, slot, [bufs] if isinstance(bufs, torch.Tensor) else list(bufs))

    def shareable_matrix(self, shape, dtype, device):
        class _Plain:                       # same process: nothing to map
            def __init__(self):
                self.tensor, self.shape, self.dtype = torch.zeros(shape, dtype=dtype, device=device), tuple(shape), dtype
        return _Plain()

    def contiguous_matrix(self, shape, dtype, device):
        return self.shareable_matrix(shape, dtype, device)          # host memory: nothing to be contiguous about

    def _spmm_rows(self, rowptr, colidx, P, rows_sel, row0, Z_old, X, gamma, Z_new, d, mirror=None):
        rp = _np(rowptr)
        acc = P.dtype
        total = 0.0
        for r in rows_sel:
            a, b = int(rp[r]), int(rp[r + 1])
            own = Z_old[row0 + r, :d]
            if b > a:
                agg = (P[a:b].unsqueeze(1) * Z_old[colidx[a:b].long(), :d].to(acc)).sum(0)
                new = (X[r, :d].to(acc) + gamma * agg).to(Z_new.dtype)
            else:
                new = own.clone()
            Z_new[r, :d] = new
            if mirror is not None:
                mp, ms, mbs = mirror
                for s in ms[int(mp[r]):int(mp[r + 1])].tolist():
                    mbs[s >> 28][s & ((1 << 28) - 1), :d] = new
            total += float((new.to(acc) - own.to(acc)).abs().sum())
        return total

    def spmm_update(self, rowptr, colidx, P, nrows, row0, Z_old, X, gamma, Z_new, d, long_threshold, partials,
                    sinks_untouched=False, mirror=None, beyond_cache=False):
        rp = _np(rowptr[:nrows + 1])
        deg = np.diff(rp)
        sel = [r for r in range(nrows) if not (long_threshold > 0 and deg[r] > long_threshold)
               and not (sinks_untouched and deg[r] == 0)]
        n = self.spmm_partials_len(nrows, 0)
        partials[:n] = 0
        partials[0] = self._spmm_rows(rowptr, colidx, P, sel, row0, Z_old, X, gamma, Z_new, d, mirror)

    def spmm_update_long(self, rowptr, colidx, P, long_rows, waves_per_row, row0, Z_old, X, gamma, Z_new, d,
                         partials, mirror=None):
        assert waves_per_row in (4, 16)
        for i, r in enumerate(long_rows.tolist()):
            partials[i] = self._spmm_rows(rowptr, colidx, P, [r], row0, Z_old, X, gamma, Z_new, d, mirror)

    def spmm_split_slab_len(self, n_segments, d):
        return n_segments * (-(-d // 8) * 8)

    def spmm_update_split(self, rowptr, colidx, P, split_rows, seg_ptr, seg_row, edges_per_segment, row0, Z_old, X,
                          gamma, Z_new, d, slab, partials, mirror=None):
        assert edges_per_segment % 64 == 0 and seg_row.numel() == int(seg_ptr[-1])
        for i, r in enumerate(split_rows.tolist()):
            partials[i] = self._spmm_rows(rowptr, colidx, P, [r], row0, Z_old, X, gamma, Z_new, d, mirror)

    def spmm_class_slab_len(self, n_slots, d):
        return n_slots * (-(-d // 8) * 8)

    def spmm_update_class(self, colidx, P, item_e0, item_len, item_slot, items_per_block, class_rows, slot_ptr, row0,
                          Z_old, X, gamma, Z_new, d, slab, partials, mirror=None, beyond_cache=False):
        """The XCD-affine pass: partial sums per item into the slab, a row's slots added in order.  Also checks the
        layout contract the kernel relies on: whole blocks, every item of block w gathers only rows of XCD class w % 8."""
        e0, ln, sl = _np(item_e0), _np(item_len), _np(item_slot)
        assert 4 <= items_per_block <= 64 and e0.size % items_per_block == 0 and e0.size // items_per_block % 8 == 0
        acc = P.dtype
        ld = -(-d // 8) * 8
        view = slab[:self.spmm_class_slab_len(int(slot_ptr[-1]), d)].view(-1, ld)
        view.fill_(float("nan"))
        for k in range(e0.size):
            if ln[k] == 0:
                assert sl[k] == -1
                continue
            a, b = int(e0[k]), int(e0[k]) + int(ln[k])
            cols = colidx[a:b].long()
            assert bool((xcd_class(cols) == (k // items_per_block) % 8).all()), "item gathers a row of another XCD class"
            view[sl[k], :d] = (P[a:b].unsqueeze(1) * Z_old[cols, :d].to(acc)).sum(0)
        mp = None if mirror is None else mirror
        for i, r in enumerate(class_rows.tolist()):
            agg = torch.zeros(d, dtype=acc)
            for s_ in range(int(slot_ptr[i]), int(slot_ptr[i + 1])):
                agg = agg + view[s_, :d]
            own = Z_old[row0 + r, :d]
            new = (X[r, :d].to(acc) + gamma * agg).to(Z_new.dtype)
            Z_new[r, :d] = new
            if mp is not None:
                p_, ms, mbs = mp
                for s_ in ms[int(p_[r]):int(p_[r + 1])].tolist():
                    mbs[s_ >> 28][s_ & ((1 << 28) - 1), :d] = new
            partials[i] = float((new.to(acc) - own.to(acc)).abs().sum())

    def reduce_partials(self, partials, n, ws, out):
        out[0] = partials[:n].sum()

    def l1_distance(self, A, B, d, ws, out, sq_a=None):
        acc = torch.float64 if A.dtype == torch.float64 else torch.float32
        out[0] = (A[:, :d].to(acc) - B[:, :d].to(acc)).abs().sum().double()
        if sq_a is not None:
            sq_a[:A.shape[0]] = A[:, :d].to(acc).pow(2).sum(1)

    def gather_rows(self, src, idx, d, dst):
        dst[:idx.numel(), :d] = src[idx.long(), :d]

    def pair_cosine(self, A, B, d, out, ws):
        out.copy_(O.cosine_similarity(A[:, :d], B[:, :d]))
