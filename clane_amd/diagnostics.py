"""What a measurement needs to know about a SweepEngine: per-kernel times, the algorithmic bytes behind them (SURVEY 8d
gather model), the kernel configuration a PMC measurement is valid for, bytes exchanged per sweep."""
from __future__ import annotations

import numpy as np

from .plan import lanes_per_row


class DiagnosticsMixin:
    """``SweepEngine``'s read-only views for bench.py and the profiling tools."""

    def kernel_times_ms(self):
        """{'split','hub','mid','main'} -> ms per SWEEP (summed over the blocks, averaged over the recorded
        sweeps); call after a synchronize.  Event order per block: 0 start, 4 after split, 1 after hub, 2 after mid,
        3 after main."""
        t = np.array([(e0.elapsed_time(e4), e4.elapsed_time(e1), e1.elapsed_time(e2), e2.elapsed_time(e3))
                      for _, e0, e1, e2, e3, e4 in self.kernel_events]).reshape(-1, 4)
        self.kernel_events = []
        if not len(t):
            return {}
        per_sweep = t.reshape(-1, len(self.blocks), 4).sum(1).mean(0)
        return dict(zip(("split", "hub", "mid", "main"), per_sweep.tolist()))

    def collective_times_ms(self):
        """{'exchange_exposed', 'allreduce'} -> ms per sweep seen from the sweep's stream (averaged over the recorded
        sweeps; call after a synchronize): how long it waited for the row exchange after its own kernels were done, and
        for the all-reduce of the delta.  Empty when nothing was recorded (one GPU, or time_collectives off)."""
        ev, self.collective_events = self.collective_events, []
        if not ev:
            return {}
        t = np.array([(a.elapsed_time(b), b.elapsed_time(c)) for a, b, c in ev])
        return {"exchange_exposed": float(t[:, 0].mean()), "allreduce": float(t[:, 1].mean()), "sweeps_timed": len(ev)}

    def kernel_bytes(self):
        """Algorithmic bytes per SWEEP of each K3 kernel (SURVEY.md section 8d gather model, split by the
        rows each kernel owns): per row  deg*(d*s + 4 + sizeof P) + 3*d*s + 8;  rows without out-edges: 8."""
        s, ps = self.Zcur.element_size(), self.P.element_size()
        deg = np.diff(self.local.rowptr)
        per_row = deg * (self.d * s + 4 + ps) + np.where(deg > 0, 3 * self.d * s, 0) + 8   # sinks: rowptr only
        is_long = deg > self.long_threshold if self.long_threshold > 0 else np.zeros_like(deg, dtype=bool)
        is_class = deg > self.class_threshold if self.class_threshold > 0 else np.zeros_like(deg, dtype=bool)
        is_long = is_long | is_class
        is_split = (is_long & (deg > self.split_edges) if self.split_edges > 0 else np.zeros_like(is_long)) & ~is_class
        is_hub = is_long & (deg > self.hub_threshold) & ~is_split & ~is_class
        return {"main": int(per_row[~is_long].sum()) + 8,
                "mid": int(per_row[is_long & ~is_hub & ~is_split & ~is_class].sum()),
                "hub": int(per_row[is_hub].sum()), "split": int(per_row[is_split | is_class].sum())}

    def launches_per_sweep(self) -> int:
        """Launches of each K3 kernel per sweep: one per launch block and column tile."""
        return len(self.blocks) * len(self.tiles)

    def kernel_names(self):
        """Names of the K3 kernels behind the keys of kernel_times_ms() / kernel_bytes()."""
        narrow = self.d > 0 and lanes_per_row(self.d_plan if len(self.tiles) > 1 else self.d, self.dtype) < 64
        return {"main": "spmm_update_subrow_kernel" if narrow else "spmm_update_kernel",
                "mid": "spmm_long_kernel<4 waves>", "hub": "spmm_long_kernel<16 waves>",
                "split": "spmm_class_chunk_kernel+combine" if self.class_threshold > 0
                else "spmm_split_segment_kernel+combine"}

    def estimated_sweep_seconds(self) -> float:
        """Rough time of one sweep on this division, the SAME number on every rank (it feeds decisions all ranks
        must take alike, e.g. whether the host check lags one sweep): the whole graph's gather-model bytes / ranks
        at the HBM peak."""
        s = self.Zcur.element_size()
        d = self.d_full
        total = self.E_total * (d * s + 8) + self.V * 3 * d * s
        return total / max(self.world, 1) / 8e12

    def kernel_config(self) -> dict:
        """Everything that decides which K3 kernels a sweep launches over which rows, and with which compile-time
        tuning: measurements of a kernel (profiles/traffic.json) are only valid for the configuration they were
        taken with, and bench.py refuses to quote them for another."""
        return {"build": self.k.build_info(), "dtype": str(self.dtype).replace("torch.", ""), "d": self.d,
                "lanes_per_row": lanes_per_row(self.d, self.dtype) if self.d > 0 else 0,
                "rows": int(self.part.n_local), "edges": int(self.E_loc), "launch_blocks": len(self.blocks),
                "long_threshold": self.long_threshold, "score_threshold": self.score_threshold,
                "hub_threshold": self.hub_threshold,
                "split_edges": self.split_edges, "segment_edges": self.segment_edges,
                "class_threshold": self.class_threshold, "class_chunk": self.class_chunk, "class_k1": self.class_k1,
                "class_phases": self.class_phases, "phase_threshold": self.phase_threshold,
                "class_of_row": "xor-fold of 3-bit groups" if self.class_threshold else None,
                "class_affinity": self.class_affinity, "mega_segment_edges": self.mega_segment_edges,
                "class_items_per_block": [c[6] for c in self.class_rows if c is not None][:1],
                "column_tiles": len(self.tiles),
                # CLANE_SPMM_TABLE_BEYOND_CACHE changes the instances only where two fp32 rows share an instruction
                "fewer_loads_in_flight": bool(self.beyond_cache and self.d > 0 and (str(self.dtype), lanes_per_row(
                    self.d_plan if len(self.tiles) > 1 else self.d, self.dtype)) in (("torch.float32", 32), ("torch.bfloat16", 16))),
                "hot_rows_first": self.hot_rows_first, "exchange": self.exchange,
                }

    def exchange_bytes_per_sweep(self) -> int:
        """Bytes this rank RECEIVES per sweep (all-gather of the live spans)."""
        s = self.Zcur.element_size()
        if self.halo:
            return self.part.recv_rows_per_sweep() * self.ld * s
        if self.columns:
            return 0
        return sum((b.span[1] - b.span[0] - b.nrows) * self.ld * s for b in self.blocks if b.span is not None)
