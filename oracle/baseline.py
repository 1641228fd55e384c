"""The CPU legs of bench.py -- TEST INFRASTRUCTURE, like everything under oracle/: the oracle's first sweep (the checker
of the GPU's first sweep) and the oracle timed on the host cores of the box the benchmark runs on (`cpu_baseline`: the
plain-C restatement; `cpu_baseline_torch`: the PyTorch-CPU restatement SURVEY 8d names).  Only bench.py's parity /
cpu_baseline leg and tests/ import this module; nothing here is ever the thing measured or shipped."""
from __future__ import annotations

import time

import numpy as np
import torch


def oracle_first_sweep(csr, X, P_host, gamma):
    """The C oracle's first sweep from Z = X (oracle/clane_oracle.c); P from the oracle's own build_P
    (graph.py:118-128) unless one is given.  Returns (Z1, seconds of the sweep, threads, the P used, X as fp32)."""
    from oracle import clane_oracle_c as OC
    Xf = X.float() if X.dtype != torch.float32 else X    # the oracle computes in fp32 on the (bf16-)rounded inputs
    if P_host is None:
        P_host, _ = OC.build_P(csr.rowptr, csr.colidx, Xf)
    out = torch.empty_like(Xf)
    t0 = time.perf_counter()
    Z, _ = OC.sweep(csr.rowptr, csr.colidx, P_host.float(), Xf, Xf, gamma, out=out)
    return Z, time.perf_counter() - t0, OC.threads(), P_host.float(), Xf


def cpu_baseline(csr, Xf, P_host, gamma, Z1_oracle, first, budget_s=8.0):
    """The oracle's sweep timed on this box's host cores: the plain-C restatement (oracle/clane_oracle.c,
    OpenMP over rows, same CSR / fp32) -- kind "port"."""
    from oracle import clane_oracle_c as OC
    threads = OC.threads()
    n = int(max(1, min(20, budget_s // max(first, 1e-3))))
    Za, Zb = Z1_oracle.clone(), torch.empty_like(Z1_oracle)
    t0 = time.perf_counter()
    for _ in range(n):
        Zb, _ = OC.sweep(csr.rowptr, csr.colidx, P_host, Xf, Za, gamma, out=Zb)
        Za, Zb = Zb, Za
    per = (time.perf_counter() - t0) / n
    return {"value": 1.0 / per, "unit": "sweeps/s", "cores": threads, "kind": "port",
            "sample": f"{n} full sweeps of the same graph by oracle/clane_oracle.c (plain C, OpenMP over rows, "
                      f"{threads} threads), P from the oracle's own build_P (graph.py:118-128)"}


def cpu_baseline_torch(csr, Xf, P_host, gamma, budget_s=8.0):
    """The PyTorch-CPU restatement SURVEY 8d names --  Z = X + gamma * (P @ Z)  plus the L1 delta, as in
    oracle/clane_oracle.py:sweep -- with P as a torch.sparse_csr_tensor (its CPU kernel is parallel over rows; the COO
    form of torch.sparse.mm is not: 0.127 sweeps/s on 128 threads against 0.130 on one in round 2), on all host
    threads and on ONE thread.  FULL sweeps (1 warm-up + up to 10 timed, at least one) whenever the estimate of one sweep
    fits the budget (config 3 on all threads: ~4 s a sweep -- it does; SURVEY 8d asks for full sweeps); else a bounded
    SAMPLE of the workload, a seeded random 1/m of the rows (same degree mix; every m-th row would not do: R-MAT's hubs
    sit on the ids with trailing zero bits), scaled by the share of the edges the sample holds -- `sample` says which."""
    import warnings
    deg = np.diff(csr.rowptr)
    E, V = int(csr.rowptr[-1]), csr.num_vertices
    Z = Xf
    all_threads = torch.get_num_threads()

    def timed(stride, threads, max_reps):
        rows = np.arange(V, dtype=np.int64) if stride == 1 else \
            np.sort(np.random.default_rng(stride).choice(V, size=max(1, V // stride), replace=False))
        if stride == 1:
            crow, cols, vals = csr.rowptr, csr.colidx, P_host
        else:
            take = np.repeat(csr.rowptr[rows], deg[rows]) + (np.arange(int(deg[rows].sum())) -
                                                             np.repeat(np.cumsum(deg[rows]) - deg[rows], deg[rows]))
            crow = np.concatenate([[0], np.cumsum(deg[rows])])
            cols, vals = csr.colidx[take], P_host[torch.from_numpy(take)]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                    # "Sparse CSR tensor support is in beta state"
            Ps = torch.sparse_csr_tensor(torch.from_numpy(np.asarray(crow, dtype=np.int64)),
                                         torch.from_numpy(np.asarray(cols).astype(np.int64)), vals, size=(rows.size, V))
        rows_t = torch.from_numpy(rows)
        Xs, Zs = (Xf, Z) if stride == 1 else (Xf[rows_t], Z[rows_t])
        sink = torch.from_numpy(deg[rows] == 0)
        n_edges = int(deg[rows].sum())
        torch.set_num_threads(threads)
        try:
            times = []
            spent = 0.0
            for i in range(max_reps + 1):                      # first pass = warm-up
                t0 = time.perf_counter()
                Zn = Xs + gamma * (Ps @ Z)
                Zn[sink] = Zs[sink]
                (Zn - Zs).abs().sum()
                dt = time.perf_counter() - t0
                spent += dt
                if i:
                    times.append(dt)
                if i >= 1 and spent + dt > budget_s:
                    break
        finally:
            torch.set_num_threads(all_threads)
        share = n_edges / max(E, 1)
        return min(times) / share, share, rows.size, len(times)

    def figure(threads, probe_rows):
        probe, _, _, _ = timed(max(1, V // probe_rows), threads, 1)           # a small probe sizes the sample
        # the probe (a small sample, scaled) over-estimates: its per-call costs are scaled too.  Full sweeps whenever
        # the ESTIMATE of one fits the budget (config 3, all threads: estimated 4-7 s, really 3.5 s)
        stride = 1 if probe <= budget_s else max(2, int(np.ceil(probe * 3 / budget_s)))
        per, share, n_rows, reps = timed(stride, threads, 10 if stride == 1 else 2)
        if stride == 1:
            return per, f"{reps} full sweeps (after 1 warm-up), best"
        # A sample's time is not proportional to its edges alone (per-call costs that do not shrink with the sample --
        # thread wake-ups, touching the whole of Z: seconds at 16M vertices -- would be multiplied by 1/share): grow the
        # sample until it takes a real share of the budget, then take the line through the two largest samples,
        # cost(E) = a + b * edges.
        samples = [(share, per * share, n_rows, stride)]
        while samples[-1][1] < budget_s / 6 and stride > 2 and len(samples) < 4:
            stride = max(2, stride // 4)
            per_n, share_n, rows_n, _ = timed(stride, threads, 2)
            samples.append((share_n, per_n * share_n, rows_n, stride))
        if len(samples) == 1:               # the first sample was big enough: a second one of half the size for the line
            per_n, share_n, rows_n, _ = timed(2 * stride, threads, 2)
            samples.insert(0, (share_n, per_n * share_n, rows_n, 2 * stride))
        (s0, t0, _, _), (s1, t1, r1, st1) = samples[-2], samples[-1]
        if t1 > t0 and s1 > s0:
            whole = t1 + (t1 - t0) / (s1 - s0) * (1.0 - s1)
            how = "the line through the two largest samples (fixed per-call cost + per-edge cost)"
        else:                               # the per-call cost drowns the difference: no slope to extend
            whole = max(t1, t0)
            how = ("NOTHING: the two samples took the same time (a per-call cost that does not shrink with the sample), so "
                   "this is a LOWER bound of a sweep's time, i.e. an upper bound of sweeps/s")
        return whole, (f"random row samples, the largest 1/{st1} of the rows ({r1} rows, {s1:.1%} of the edges), best of 2 "
                       f"each, extended to a whole sweep by {how}")

    per_all, what_all = figure(all_threads, 20_000)
    per_1, what_1 = figure(1, 5_000)
    return {"value": 1.0 / per_all, "unit": "sweeps/s", "cores": all_threads, "kind": "port",
            "sample": f"X + gamma*(P @ Z) + L1 delta with P a torch.sparse_csr_tensor (oracle/clane_oracle.py:sweep): "
                      f"{what_all}; torch {torch.__version__}, {all_threads} threads",
            "one_thread": {"value": 1.0 / per_1, "unit": "sweeps/s", "cores": 1,
                           "sample": f"the same with torch.set_num_threads(1): {what_1}"}}
