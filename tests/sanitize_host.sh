#!/bin/bash
# Sanitizers over the host-side native code, CPU only (GPU ASAN is not available on this pool):
#   1. AddressSanitizer + UBSan: the V/E parser (csrc/host_loader.cpp, up to 16 threads) and the C oracle (OpenMP);
#   2. ThreadSanitizer: the parser's threads (the oracle is left out of that run: libgomp is not instrumented).
# Usage: tests/sanitize_host.sh   (exit 0 and two "sanitize_host: ok" lines when clean)
set -euo pipefail
cd "$(dirname "$0")/.."
out=$(mktemp -d)
trap 'rm -rf "$out"' EXIT
mkdir "$out/data"
CXX="g++ -O1 -g -std=c++17 -pthread -fno-omit-frame-pointer"

gcc -c -O1 -g -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer oracle/clane_oracle.c -o "$out/oracle.o"
$CXX -fopenmp -fsanitize=address,undefined -fno-sanitize-recover=undefined \
    tests/sanitize_host_main.cpp clane_amd/csrc/host_loader.cpp "$out/oracle.o" -o "$out/sanitize_host"
ASAN_OPTIONS=detect_leaks=1 "$out/sanitize_host" "$out/data"

gcc -c -O1 -g -fopenmp oracle/clane_oracle.c -o "$out/oracle_plain.o"
$CXX -fopenmp -fsanitize=thread tests/sanitize_host_main.cpp clane_amd/csrc/host_loader.cpp "$out/oracle_plain.o" \
    -o "$out/sanitize_host_tsan"
"$out/sanitize_host_tsan" "$out/data" loader-only
