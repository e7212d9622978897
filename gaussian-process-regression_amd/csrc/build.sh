#!/usr/bin/env bash
# Builds libgprc_native$SUFFIX.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
# -amdgpu-mfma-vgpr-form keeps MFMA accumulators in VGPRs: without it hipcc bounces the 128
# accumulator registers through AGPRs around every loop iteration (measured 36 vs 77 TFLOP/s).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../lib"
mkdir -p "$out"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
NBW="${GPRC_NB:-512}"
SUFFIX="${GPRC_LIB_SUFFIX:-}"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -mllvm -amdgpu-mfma-vgpr-form=1 -Wall -Wno-unused-function -DGPRC_NB="$NBW" ${GPRC_EXTRA_FLAGS:-})
objs=()
pids=()
for src in gprc_api gprc_mgpu kernels_fill kernels_chol kernels_vec kernels_eig; do
  rm -f "$out/$src$SUFFIX.o"                     # a failed compile must not link last time's object
  "$HIPCC" "${FLAGS[@]}" -c "$here/$src.hip" -o "$out/$src$SUFFIX.o" &
  pids+=($!)
  objs+=("$out/$src$SUFFIX.o")
done
for pid in "${pids[@]}"; do wait "$pid"; done    # (a bare `wait` returns 0 whatever the jobs did)
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$out/libgprc_native$SUFFIX.so" "${objs[@]}" -ldl -lpthread
echo "built $out/libgprc_native$SUFFIX.so"
