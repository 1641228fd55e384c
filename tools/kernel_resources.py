#!/usr/bin/env python3
"""Registers / occupancy / LDS of every kernel in libclane_hip.so, from hipcc's -Rpass-analysis=kernel-resource-usage
(cross-compiles without a GPU).  Usage: python tools/kernel_resources.py [substring ...]"""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-shared",
       "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/clane_resources.so", str(ROOT / "clane_amd/csrc/clane_abi.hip")]
cmd += [a for a in sys.argv[1:] if a.startswith("-D")]
want = [a for a in sys.argv[1:] if not a.startswith("-D")]
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
names, rows = [], []
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0].split(" [")[0]
    g = lambda k: re.search(k + r": (\d+)", b).group(1)   # noqa: E731
    names.append(name)
    rows.append((g("VGPRs"), g("AGPRs"), g("SGPRs"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]"),
                 g(r"ScratchSize \[bytes/lane\]")))
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
print(f"{'VGPR':>5}{'AGPR':>5}{'SGPR':>5}{'occ':>4}{'LDS':>7}{'scr':>5}  kernel")
for n, r in zip(dem, rows):
    short = re.sub(r"^void clane::", "", n)
    if not want or any(w in short for w in want):
        print(f"{r[0]:>5}{r[1]:>5}{r[2]:>5}{r[3]:>4}{r[4]:>7}{r[5]:>5}  {short[:140]}")
