#!/usr/bin/env bash
# Build the reference's own CPU path into oracle/_ref/ (test infrastructure).
#
# Compiles the reference's C sources WHERE THEY LIE under $REF/src together
# with oracle/ref_harness.c.  Nothing is copied, nothing is stubbed: the
# CUDA-facing wrappers in csr.c/hll.c are unreferenced by the harness and are
# discarded at link time (--gc-sections), so their undefined cuda symbols
# never need resolving.  Outputs (git-ignored, but they DO travel to the GPU
# box with gpurun):
#   oracle/_ref/ref_strict  -O2, strict IEEE: generates the golden vectors
#   oracle/_ref/ref_dropin  the reference's unmodified driver + host layer
#                           linked against libspmv_scpa_amd.so (drop-in proof)
#   oracle/_ref/ref_dropin_cached  the same + oracle/seam_cache_on.c (the
#                           opt-in resident cache of the seam switched on)
#   oracle/_ref/ref_fast    the reference's flags (CMakeLists.txt:11-18:
#                           -O3 -fopenmp -ffast-math -funroll-loops) with
#                           -march=x86-64-v3 instead of -march=native, since
#                           the binary is built here and run on the GPU
#                           box's (different) host CPU: the CPU baseline.
# Skips quietly when the reference tree is absent (the GPU box).
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
REF="${SPMV_REFERENCE_DIR:-/root/reference}"
out="$here/_ref"
if [ ! -d "$REF/src" ]; then
    echo "build_ref: $REF not present; keeping prebuilt $out" >&2
    exit 0
fi
mkdir -p "$out"
srcs=("$REF/src/mmio.c" "$REF/src/utils.c" "$REF/src/vector.c"
      "$REF/src/csr.c" "$REF/src/hll.c")
common=(-std=c99 -D_GNU_SOURCE -fopenmp -I"$REF/include"
        -ffunction-sections -fdata-sections -Wl,--gc-sections -w)
gcc "${common[@]}" -O2 "$here/ref_harness.c" "${srcs[@]}" -lm \
    -o "$out/ref_strict"
gcc "${common[@]}" -O3 -march=x86-64-v3 -ffast-math -funroll-loops \
    "$here/ref_harness.c" "${srcs[@]}" -lm -o "$out/ref_fast"
echo "build_ref: built $out/ref_strict $out/ref_fast"

# Drop-in proof: the reference's UNMODIFIED driver and host
# layer (main.c, csr.c, hll.c, ...) linked against libspmv_scpa_amd.so, which
# exports the reference's own 11 plugin symbols (include/spmv_ref_abi.h) in
# place of its CUDA translation units.  Run on the GPU box by
# tests/test_gpu_dropin.py.
lib="$here/../spmv_scpa_amd/lib"
if [ -f "$lib/libspmv_scpa_amd.so" ]; then
    gcc -std=c99 -D_GNU_SOURCE -fopenmp -I"$REF/include" -w -O3 \
        "$REF/src/main.c" "$REF/src/logger.c" "${srcs[@]}" \
        -L"$lib" -lspmv_scpa_amd -Wl,-rpath,'$ORIGIN/../../spmv_scpa_amd/lib' \
        -Wl,-rpath,/opt/rocm/lib -lm -o "$out/ref_dropin"
    echo "build_ref: built $out/ref_dropin (reference driver + MI355X kernels)"
    # the same, with the opt-in "keep the last upload" of the seam switched on
    # from a constructor (oracle/seam_cache_on.c: the reference's sources stay
    # unmodified); tests/test_gpu_dropin.py compares the two runs
    gcc -std=c99 -D_GNU_SOURCE -fopenmp -I"$REF/include" -w -O3 \
        "$REF/src/main.c" "$REF/src/logger.c" "${srcs[@]}" \
        "$here/seam_cache_on.c" \
        -L"$lib" -lspmv_scpa_amd -Wl,-rpath,'$ORIGIN/../../spmv_scpa_amd/lib' \
        -Wl,-rpath,/opt/rocm/lib -lm -o "$out/ref_dropin_cached"
fi
