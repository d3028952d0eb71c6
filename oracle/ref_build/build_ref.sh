#!/bin/bash
# build_ref.sh -- builds the REFERENCE's own Gaussian-surfel rasterizer for MI355X as a test checker:
#   /root/reference/submodules/diff-gaussian-rasterization/cuda_rasterizer/*.{cu,h}  (+ its vendored third_party/glm)
#   -> oracle/_ref/libref_rasterizer.so
# The sources are compiled from where they lie: hipify-perl (ROCm's own CUDA->HIP source translator, /opt/rocm/bin) writes
# the translated units into a scratch directory under oracle/_ref/ that is deleted after linking; nothing of the reference is
# kept in the repository and oracle/_ref/ is git-ignored.  No header, library or tool is stood in for: the only edits are
# token-level fixes of what hipify-perl leaves behind (three CUDA-only include lines it cannot map, and the `<< <` / `>> >`
# launch brackets it does not recognise), and `__trap` -> `__builtin_trap` on the command line.
# -ffp-contract=off: every fp32 expression is evaluated as the reference wrote it (no compiler-chosen FMA contraction, which
# differs between nvcc and clang and flips last bits of the depth keys); this is the evaluation the oracle restates.
# The torch binding (rasterize_points.cu / ext.cpp) is not built; ref_shim.hip calls CudaRasterizer::Rasterizer directly.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${SOAR_REFERENCE:-/root/reference}/submodules/diff-gaussian-rasterization"
OUT="$HERE/../_ref"
[ -d "$REF/cuda_rasterizer" ] || { echo "reference sources not found at $REF (prebuilt $OUT is used as is)"; exit 3; }
mkdir -p "$OUT"
TMP="$(mktemp -d "$OUT/hipified.XXXXXX")"
trap 'rm -rf "$TMP"' EXIT
for f in "$REF"/cuda_rasterizer/*.cu "$REF"/cuda_rasterizer/*.h; do
    /opt/rocm/bin/hipify-perl "$f" 2>/dev/null | sed -e '/#include ""/d' -e '/cub\/device\/device_radix_sort.cuh/d' \
        -e '/cooperative_groups\/reduce.h/d' -e 's/<< </<<</g' -e 's/>> >/>>>/g' > "$TMP/$(basename "$f")"
done
# REF_VARIANT=fast: the compiler's defaults (-O3, FMA contraction on) -> libref_rasterizer_fast.so, used only by
# tests/tools/ref_compare.py to time the reference's kernels without the parity build's floating-point restriction.
NAME=libref_rasterizer.so; FP="-O2 -ffp-contract=off"
if [ "${REF_VARIANT:-}" = fast ]; then NAME=libref_rasterizer_fast.so; FP="-O3"; fi
FLAGS="--offload-arch=gfx950 $FP -fPIC -std=c++17 -w -D__trap=__builtin_trap -I$REF/third_party/glm -I$TMP"
pids=()
for u in forward backward rasterizer_impl; do
    hipcc $FLAGS -x hip -c "$TMP/$u.cu" -o "$TMP/$u.o" & pids+=($!)
done
hipcc $FLAGS -c "$HERE/ref_shim.hip" -o "$TMP/ref_shim.o" & pids+=($!)
for p in "${pids[@]}"; do wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/$NAME" "$TMP"/forward.o "$TMP"/backward.o "$TMP"/rasterizer_impl.o "$TMP"/ref_shim.o
echo "built $OUT/$NAME"
