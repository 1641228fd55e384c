#!/bin/bash
# AddressSanitizer + UBSan over the host-side native code (the V/E parser and the C oracle), CPU only -- GPU ASAN is
# not available on this pool.  Usage: tests/sanitize_host.sh   (exit 0 and "sanitize_host: ok" when clean)
set -euo pipefail
cd "$(dirname "$0")/.."
out=$(mktemp -d)
trap 'rm -rf "$out"' EXIT
gcc -c -O1 -g -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer oracle/clane_oracle.c -o "$out/oracle.o"
g++ -O1 -g -std=c++17 -pthread -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined \
    tests/sanitize_host_main.cpp clane_amd/csrc/host_loader.cpp "$out/oracle.o" -o "$out/sanitize_host"
mkdir "$out/data"
ASAN_OPTIONS=detect_leaks=1 "$out/sanitize_host" "$out/data"
