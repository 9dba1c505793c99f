#!/bin/bash
# Builds timing-only variants of the library with one ingredient of the fused NLTV kernel removed (wrong results by
# construction) into tools/_whatif/ (git-ignored; travels to the GPU box).  Run there:
#   for f in tools/_whatif/libnf_*.so; do FALDOI_LIB=$f python tools/time_nltv.py | tail -1; done
set -e
cd "$(dirname "$0")/.."
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize -mllvm -disable-promote-alloca-to-vector"
build() { /opt/rocm/bin/hipcc $FLAGS -DFALDOI_NFWHATIF=$1 -Iinclude -Ifaldoi-ipol_amd/csrc -o tools/_whatif/libnf_$1.so faldoi-ipol_amd/csrc/faldoi_hip.hip 2>/dev/null; echo built $1; }
for w in "$@"; do build $w & if (( $(jobs -r | wc -l) >= 4 )); then wait -n; fi; done
wait
