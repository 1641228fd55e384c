"""bench.py's host logic without a GPU: the self-launch of N > 1 ranks, the stale-traffic rule of the roofline
line, and the sampled PyTorch-CPU baseline."""
import json
import os
import subprocess
import sys
from pathlib import Path
from types import SimpleNamespace

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def test_gpus_flag_without_launcher_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with WORLD_SIZE unset (the driver's command form) must not die on the flag: it
    starts two fresh rank processes.  Without a GPU those ranks refuse to run (no CPU fallback) and the exit
    code comes back; with one, the rehearsal flags let both share it and one JSON line comes out."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    run = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--workload", "tiny", "--steps", "1",
                          "--warmup", "0", "--backend", "gloo", "--share-gpu", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert "starting ranks" in run.stderr and "--nproc-per-node=2" in run.stderr
    assert "the launcher and the flag disagree" not in run.stderr
    if torch.cuda.is_available():
        assert run.returncode == 0, run.stderr[-2000:]
        lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2
    else:
        assert run.returncode != 0 and "no CPU fallback" in run.stderr


def _fake_engine(**over):
    cfg = {"build": "arch=gfx950;SPMM_U=8", "dtype": "float32", "d": 256, "lanes_per_row": 64, "rows": 10, "edges": 99,
           "launch_blocks": 1, "long_threshold": 32, "hub_threshold": 32, "split_edges": 4096, "segment_edges": 4096,
           "hot_rows_first": True, "exchange": "none"}
    cfg.update(over)
    return SimpleNamespace(kernel_config=lambda: dict(cfg), columns=False), cfg


def test_traffic_is_only_quoted_for_the_configuration_it_was_measured_with(tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", tmp_path)
    (tmp_path / "profiles").mkdir()
    eng, cfg = _fake_engine()
    table = {"w_n1": {"source": "profiles/x.md", "kernel_config": cfg, "k": {"bytes_per_launch": 123.0, "avg_us_under_pmc": 7.5}},
             "old_n1": {"k": {"bytes_per_launch": 5.0}}}
    (tmp_path / "profiles" / "traffic.json").write_text(json.dumps(table))
    assert bench.traffic_entry("w", 1, eng, "k") == (123.0, {"source": "profiles/x.md", "avg_us_under_pmc": 7.5})
    got, why = bench.traffic_entry("w", 1, eng, "other")
    assert got is None and "not in the measurement" in why
    got, why = bench.traffic_entry("old", 1, eng, "k")                  # an entry from before configurations were kept
    assert got is None and "stale" in why
    got, why = bench.traffic_entry("nope", 1, eng, "k")
    assert got is None and "no PMC measurement" in why
    for change in ({"long_threshold": 64}, {"build": "arch=gfx950;SPMM_U=4"}, {"segment_edges": 1024}, {"d": 128}):
        other, _ = _fake_engine(**change)
        got, why = bench.traffic_entry("w", 1, other, "k")
        assert got is None and "stale" in why and list(change)[0] in why
    # the committed table: the headline workload's entry carries the configuration it was measured with
    real = json.loads((ROOT / "profiles" / "traffic.json").read_text())
    assert "kernel_config" in real["rmat2m_n1"] and real["rmat2m_n1"]["kernel_config"]["d"] == 256
    assert real["rmat2m_n1"]["spmm_class_chunk_kernel+combine"]["avg_us_under_pmc"] > 1000   # the PMC run's own kernel time


def test_torch_baseline_sampled_and_full(monkeypatch):
    """The PyTorch-CPU baseline (P as a sparse CSR tensor, whose CPU kernel threads over rows): with a tiny budget it
    times a random row sample and scales it; with room it times FULL sweeps; either way it is the same quantity as a
    plain full sweep, and the all-thread figure is not below the one-thread figure (round 2's COO form did not scale)."""
    import time
    import warnings
    import numpy as np
    from clane_amd import synth
    from oracle import baseline as B
    from oracle import clane_oracle as O
    csr = synth.rmat_csr(60_000, 1_500_000, seed=1, device="cpu")
    X = synth.gaussian_X(60_000, 64, seed=2)
    P = O.build_P_values(csr.rowptr, csr.colidx, X)
    out = B.cpu_baseline_torch(csr, X, P, 0.76, budget_s=0.001)     # tiny budget: forces row sampling
    assert out["kind"] == "port" and out["cores"] == torch.get_num_threads() and out["one_thread"]["cores"] == 1
    assert "sparse_csr_tensor" in out["sample"] and "random row samples, the largest 1/" in out["sample"]
    full_run = B.cpu_baseline_torch(csr, X, P, 0.76, budget_s=20.0)
    assert "full sweeps" in full_run["sample"] and "full sweeps" in full_run["one_thread"]["sample"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        Ps = torch.sparse_csr_tensor(torch.from_numpy(csr.rowptr), torch.from_numpy(csr.colidx.astype(np.int64)), P,
                                     size=(60_000, 60_000))
    sink = torch.from_numpy(np.diff(csr.rowptr) == 0)

    def sweep():
        Zn = X + 0.76 * (Ps @ X)
        Zn[sink] = X[sink]
        return (Zn - X).abs().sum()
    sweep()
    t0 = time.perf_counter()
    ref = sweep()
    full = time.perf_counter() - t0
    Zo, d_or = O.sweep(csr.rowptr, csr.colidx, P, X, X, 0.76)            # the CSR form is the oracle's sweep
    assert float(ref) == pytest.approx(float(d_or), rel=1e-5)
    assert 0.1 < (1.0 / full_run["value"]) / full < 10                  # full sweeps: the same quantity
    assert 0.02 < (1.0 / out["value"]) / full < 50                      # a millisecond's worth of samples, extended: the
    #                                                                     same order of magnitude (a loaded 8-core host)
    if torch.get_num_threads() >= 4:        # the CSR kernel threads (round 2's COO form: 0.98x); a loaded host gets a second try
        ratio = full_run["value"] / full_run["one_thread"]["value"]
        if ratio <= 1.2:
            again = B.cpu_baseline_torch(csr, X, P, 0.76, budget_s=20.0)
            ratio = max(ratio, again["value"] / again["one_thread"]["value"])
        assert ratio > 1.2, ratio
    assert torch.get_num_threads() == out["cores"]                      # thread count restored
