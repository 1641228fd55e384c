#!/bin/bash
# Compile the library with -save-temps and print, for the gather kernels, the sequence of global loads
# and vmcnt waits.  Healthy: "8 global_load_dwordx4" in one run followed by (counted) waits.
# Unhealthy (hipcc sank consumers into conditional-load blocks): load / s_waitcnt vmcnt(0) pairs.
set -e
SRC="$(cd "$(dirname "$0")/.." && pwd)/clane_amd/csrc/clane_abi.hip"
OUT=${1:-/tmp/clane_isa}; mkdir -p $OUT; cd $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -shared -save-temps \
  -Rpass-analysis=kernel-resource-usage -o lib.so "$SRC" 2> rpass.txt
S=clane_abi-hip-amdgcn-amd-amdhsa-gfx950.s
for K in $(grep -o "^_ZN5clane\(16spmm_long_kernel\|18spmm_update_kernel\|25spmm_update_subrow_kernel\|17edge_score_kernel\|24edge_score_subrow_kernel\|22edge_score_long_kernel\)[A-Za-z0-9_]*" $S | sort -u); do
  L=$(grep -n "^$K:" $S | head -1 | cut -d: -f1)
  printf "%-100s " "$(echo $K | c++filt | cut -c1-100)"
  sed -n "${L},\$p" $S | awk '/^\.Lfunc_end/{exit} {print}' | grep "global_load_dwordx4\|global_load_dword \|s_waitcnt vmcnt" |
    awk '{print $1, $2}' | sed 's/v\[[0-9:]*\],//; s/v[0-9]*,//' | uniq -c | awk '$2 ~ /global_load/ && $1 > m {m=$1} END{print "max grouped loads:", m}'
done
