#!/bin/bash
# Build A/B variants of libclane_hip.so into build/variants/ (shipped to the GPU box by gpurun).
# usage: tools/build_variants.sh name1:"-DFOO=1 -DBAR=2" name2:"..." ...
set -e
cd "$(dirname "$0")/.."
mkdir -p build/variants
pids=()
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -shared $flags \
      -o build/variants/libclane_hip_$name.so clane_amd/csrc/clane_abi.hip && echo "built $name [$flags]" ) &
  pids+=($!)
  if (( ${#pids[@]} >= 4 )); then wait "${pids[0]}"; pids=("${pids[@]:1}"); fi
done
wait
