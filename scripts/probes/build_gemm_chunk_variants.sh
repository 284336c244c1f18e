#!/bin/bash
# Ablation / stamp variants of csrc/gemm_f16c.hip as stand-alone shared objects for scripts/probes/gemm_chunk_variants.py
# (built here on the CPU box; the .so files travel with the snapshot, they are git-ignored)
set -e
cd "$(dirname "$0")/../.."
for v in ${TOCVP_GC_VARIANTS:-0 1 2 4 5 6 7}; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -Wno-comment -w -fPIC -shared -Iinclude -Itextocvp_amd/csrc \
    -DTOCVP_GC_ABLATE=$v -DTOCVP_GC_STAMP ${TOCVP_GC_DEFS} textocvp_amd/csrc/gemm_f16c.hip -o scripts/probes/gemm_chunk_v$v.so &
done
wait
ls -la scripts/probes/gemm_chunk_v*.so
