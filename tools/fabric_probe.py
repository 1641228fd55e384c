#!/usr/bin/env python3
"""What the fabric and RCCL do with the messages of the row-partitioned sweep -- measured, so that an N > 1 bench record
explains itself (nobody can iterate on an 8-GPU run: it happens once, at round end, on somebody else's node).

SURVEY 8(e) leaves ONE question open for north_star's plan (node rows of Z partitioned, one all-gather of the updated
rows per sweep, embedder.py:84-94 divided): does RCCL move the all-gather as a RING (7 x slice through one link
direction) or directly over the 7 point-to-point xGMI links of a GPU -- and what does a link really deliver?  This
probe answers it in child processes, one per rank, started by bench.py BEFORE the rank touches its GPU:

  * the children form their own process group (file rendezvous under /tmp) with NCCL_DEBUG=INFO
    NCCL_DEBUG_SUBSYS=INIT,GRAPH,TUNING written to one file per rank -- the timed run itself never logs;
  * they time, at the literal plan's message size (the rank's slice of Z): the in-place `all_gather_into_tensor`,
    grouped `isend` / `irecv` of the slice to and from EVERY peer (the direct exchange), `all_to_all_single`
    (what the halo division issues), and the 8-byte all-reduce of the delta -- max over ranks, GB/s per link;
  * rank 0's parent parses the logs (algorithm / protocol per collective and size, channel count, ring orders) and
    adds `rocm-smi --showtopo`'s link matrix.

Usage (bench.py does this): fabric_probe.run(...) in every rank; the dict comes back on rank 0.
Stand-alone child:  RANK=.. WORLD_SIZE=.. LOCAL_RANK=.. python tools/fabric_probe.py --child --dir DIR --slice-bytes B
"""
from __future__ import annotations

import argparse
import json
import os
import re
import subprocess
import sys
import time
from pathlib import Path

ALGOS = {0: "Tree", 1: "Ring", 2: "CollnetDirect", 3: "CollnetChain", 4: "NVLS", 5: "NVLSTree"}
PROTOS = {0: "LL", 1: "LL128", 2: "Simple"}
MAX_SLICE_BYTES = 1 << 30        # a probe, not a stress test: cap the message (config 3 at N = 8: 256 MB per rank)


def parse_rccl_log(text: str) -> dict:
    """Algorithm / protocol per (collective, bytes), channels, ring orders and the graph search's verdict out of an
    NCCL_DEBUG=INFO log (subsystems INIT, GRAPH, TUNING).  Tolerant: what is not there is simply absent."""
    out = {"tuning": [], "channels": None, "rings": [], "graphs": [], "init": None, "version": None}
    seen = set()
    for line in text.splitlines():
        # "AllGather: 268435456 Bytes -> Algo 1 proto 2 time 31234.5" (RCCL <= 2.2x: numbers) or
        # "AllGather: 268435456 Bytes -> Algo RING proto SIMPLE channel{Lo..Hi}={0..15}" (NCCL >= 2.24 wording: names)
        m = re.search(r"(\w+): (\d+) Bytes -> Algo (\w+) proto (\w+)(?: time ([0-9.eE+-]+))?(?: channel\{Lo\.\.Hi\}=\{(\d+)\.\.(\d+)\})?", line)
        if m:
            algo = ALGOS.get(int(m.group(3)), m.group(3)) if m.group(3).isdigit() else m.group(3).capitalize()
            proto = PROTOS.get(int(m.group(4)), m.group(4)) if m.group(4).isdigit() else \
                {"SIMPLE": "Simple"}.get(m.group(4), m.group(4))
            key = (m.group(1), int(m.group(2)), algo, proto)
            if key not in seen:
                seen.add(key)
                entry = {"collective": key[0], "bytes": key[1], "algo": algo, "proto": proto}
                if m.group(5):
                    entry["model_time_us"] = float(m.group(5))
                if m.group(6):
                    entry["channels"] = [int(m.group(6)), int(m.group(7))]
                out["tuning"].append(entry)
            continue
        m = re.search(r"(\d+) coll channels, (\d+) (?:collnet|nvls) channels.*?(\d+) p2p channels(?:, (\d+) p2p channels per peer)?", line)
        if m:
            out["channels"] = {"coll": int(m.group(1)), "p2p": int(m.group(3)),
                               "p2p_per_peer": int(m.group(4)) if m.group(4) else None}
            continue
        m = re.search(r"Channel (\d+)/(\d+) :\s+((?:\d+\s*)+)$", line)
        if m and len(out["rings"]) < 4:
            out["rings"].append({"channel": int(m.group(1)), "of": int(m.group(2)),
                                 "order": [int(x) for x in m.group(3).split()]})
            continue
        m = re.search(r"Pattern (\d+), crossNic (\d+), nChannels (\d+), bw ([0-9.]+)/([0-9.]+), type ([^\s,]+),? sameChannels (\d+)", line)
        if m and len(out["graphs"]) < 6:
            out["graphs"].append({"pattern": int(m.group(1)), "nChannels": int(m.group(3)), "bw_intra": float(m.group(4)),
                                  "bw_inter": float(m.group(5)), "type": m.group(6), "sameChannels": int(m.group(7))})
            continue
        m = re.search(r"nranks (\d+) .*Init COMPLETE", line)
        if m:
            out["init"] = {"nranks": int(m.group(1))}
            continue
        m = re.search(r"(RCCL version [^\n]*|NCCL version [^\n]*)", line)
        if m and out["version"] is None:
            out["version"] = m.group(1).strip()[:120]
    return out


def parse_showtopo(text: str) -> dict:
    """The matrices of `rocm-smi --showtopo` ("Weight between two GPUs", "Hops between two GPUs", "Link Type between two
    GPUs") as {name: [[...]]}; the raw text is kept beside them by the caller."""
    out, name, rows = {}, None, []
    for line in text.splitlines():
        head = re.match(r"=+\s*(Weight|Hops|Link Type) between two GPUs\s*=+", line.strip())
        if head:
            if name and rows:
                out[name] = rows
            name, rows = head.group(1).lower().replace(" ", "_"), []
            continue
        if name is None:
            continue
        m = re.match(r"\s*GPU(\d+)\s+(.*\S)\s*$", line)
        if m and not re.match(r"\s*GPU\d+\s+GPU\d+", line):
            rows.append(m.group(2).split())
        elif line.strip().startswith("=") and rows:
            out[name] = rows
            name, rows = None, []
    if name and rows:
        out[name] = rows
    return out


def showtopo(timeout_s: float = 20.0) -> dict:
    """`rocm-smi --showtopo` of this box (a separate program: nothing in this process touches the GPU for it)."""
    try:
        run = subprocess.run(["rocm-smi", "--showtopo"], capture_output=True, text=True, timeout=timeout_s)
        text = run.stdout
        return {"matrices": parse_showtopo(text), "raw": text[-6000:], "rc": run.returncode}
    except Exception as exc:                # noqa: BLE001 -- a diagnostic: its failure is a line in the record
        return {"error": f"{type(exc).__name__}: {exc}"}


# ---- the child: one per rank ------------------------------------------------------------------------------------------
def child(args) -> int:
    import datetime
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0"))
    root = Path(args.dir)
    n_dev = torch.cuda.device_count()
    dev = torch.device("cuda", 0 if (args.share_gpu or n_dev == 1) else local)
    torch.cuda.set_device(dev)
    kw = {"device_id": dev} if args.backend == "nccl" else {}
    dist.init_process_group(args.backend, init_method=f"file://{root / 'rendezvous'}", rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=args.timeout), **kw)
    slice_bytes = min(int(args.slice_bytes), MAX_SLICE_BYTES) // 16 * 16
    n = slice_bytes // 4
    table = torch.zeros(world * n, dtype=torch.float32, device=dev)
    mine = table[rank * n:(rank + 1) * n]
    mine.fill_(float(rank + 1))
    inbox = torch.zeros(world * n, dtype=torch.float32, device=dev)
    scalar = torch.zeros(1, dtype=torch.float64, device=dev)

    def all_gather():
        dist.all_gather_into_tensor(table, mine)

    def direct():
        ops = []
        for q in range(world):
            if q != rank:
                ops.append(dist.P2POp(dist.isend, mine, q))
                ops.append(dist.P2POp(dist.irecv, inbox[q * n:(q + 1) * n], q))
        for w in dist.batch_isend_irecv(ops):
            w.wait()

    def all_to_all():
        dist.all_to_all_single(inbox, table)

    def all_reduce8():
        dist.all_reduce(scalar)

    def timed(fn, reps):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        t = torch.tensor([(time.perf_counter() - t0) / reps], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t)

    res = {"slice_bytes": slice_bytes, "world": world, "backend": args.backend}
    tests = [("all_gather_into_tensor", all_gather, 5), ("all_reduce_8_bytes", all_reduce8, 50)]
    if world > 1:
        tests[1:1] = [("send_recv_every_peer", direct, 5), ("all_to_all_single", all_to_all, 5)]
    for name, fn, reps in tests:
        try:
            t = timed(fn, reps)
            entry = {"seconds": t}
            if name != "all_reduce_8_bytes":
                # every rank receives one slice from each of its world - 1 peers
                per_peer = slice_bytes if name != "all_to_all_single" else slice_bytes      # table is world slices: one per peer
                entry["GBps_received_per_rank"] = (world - 1) * per_peer / t / 1e9 if world > 1 else None
                entry["GBps_per_peer"] = per_peer / t / 1e9 if world > 1 else slice_bytes / t / 1e9
            res[name] = entry
        except Exception as exc:            # noqa: BLE001
            res[name] = {"error": f"{type(exc).__name__}: {exc}"}
    ok = bool(torch.all(table.view(world, n)[:, 0] == torch.arange(1, world + 1, device=dev, dtype=torch.float32)))
    res["all_gather_correct"] = ok
    if rank == 0:
        (root / "probe.json").write_text(json.dumps(res))
    dist.barrier()
    dist.destroy_process_group()
    return 0


# ---- the parent side: every rank of bench.py calls this before it initialises its GPU -----------------------------------
def run(world: int, rank: int, local_rank: int, slice_bytes: int, backend: str = "nccl", share_gpu: bool = False,
        timeout_s: float = 90.0) -> dict | None:
    """Start this rank's probe child and wait for it.  Returns the merged record on rank 0 (None elsewhere).  Never
    raises: whatever goes wrong becomes {"error": ...} -- a diagnostic must not cost the measurement."""
    t0 = time.perf_counter()
    try:        # one directory per LAUNCH, the same name in every rank: the launcher's pid and its start time
        born = Path(f"/proc/{os.getppid()}/stat").read_text().rsplit(")", 1)[1].split()[19]
    except Exception:                       # noqa: BLE001
        born = "0"
    root = Path(os.environ.get("TMPDIR", "/tmp")) / (f"clane_fabric_probe_{os.getppid()}_{born}_"
                                                    f"{os.environ.get('MASTER_PORT', '0')}")
    try:
        root.mkdir(parents=True, exist_ok=True)
        log = root / f"rccl_rank{rank}.log"
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(local_rank),
                   NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,GRAPH,TUNING", NCCL_DEBUG_FILE=str(log),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        cmd = [sys.executable, str(Path(__file__).resolve()), "--child", "--dir", str(root), "--slice-bytes", str(slice_bytes),
               "--backend", backend, "--timeout", str(int(timeout_s))] + (["--share-gpu"] if share_gpu else [])
        proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s + 30)
        if rank != 0:
            return None
        out = {"how": "child processes with their own process group (NCCL_DEBUG=INFO, INIT,GRAPH,TUNING to a file per rank), "
                      "before the timed run's ranks touch their GPUs: the timed run itself never logs",
               "child_rc": proc.returncode, "wall_s": None}
        if proc.returncode != 0:
            out["error"] = (proc.stderr or proc.stdout)[-600:]
        if (root / "probe.json").exists():
            out["microbench"] = json.loads((root / "probe.json").read_text())
            mb = out["microbench"]
            ag, sr = mb.get("all_gather_into_tensor", {}), mb.get("send_recv_every_peer", {})
            if world > 1 and "seconds" in ag and "seconds" in sr:
                # a ring moves (world - 1) slices through every link in turn; the direct exchange one slice per link
                ratio = ag["seconds"] / sr["seconds"]
                out["all_gather_vs_direct_exchange"] = {
                    "time_ratio": ratio,
                    "reading": ("the all-gather takes about as long as one slice per link: RCCL moves it directly (or a "
                                "ring at full multi-link rate)" if ratio < 2.0 else
                                f"the all-gather takes {ratio:.1f}x the direct exchange of the same slices: a ring-like "
                                f"schedule -- the engine's grouped send/recv (exchange='halo') is the faster form here")}
        if log.exists():
            text = log.read_text(errors="replace")
            out["rccl_log"] = parse_rccl_log(text)
            out["rccl_log"]["lines"] = text.count("\n")
            # the lines themselves, should a newer RCCL word them differently than parse_rccl_log expects
            keep = [ln[-220:] for ln in text.splitlines() if re.search(r"Algo|Channel \d+/|Pattern|channels|Init COMPLETE|XGMI|version", ln)]
            out["rccl_log"]["sample_lines"] = keep[:8] + keep[-8:] if len(keep) > 16 else keep
        else:
            out["rccl_log"] = None          # gloo rehearsal, or RCCL wrote nothing
        out["topology"] = showtopo()
        out["wall_s"] = time.perf_counter() - t0
        return out
    except Exception as exc:                # noqa: BLE001
        return {"error": f"{type(exc).__name__}: {exc}", "wall_s": time.perf_counter() - t0} if rank == 0 else None
    finally:
        if rank == 0:
            for f in list(root.glob("*")) if root.exists() else []:
                try:
                    if f.name == "rendezvous" or f.suffix in (".json",):
                        f.unlink()
                except OSError:
                    pass


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--dir", required=True)
    ap.add_argument("--slice-bytes", type=int, required=True)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--share-gpu", action="store_true")
    ap.add_argument("--timeout", type=int, default=150)
    a = ap.parse_args()
    sys.exit(child(a) if a.child else 2)
