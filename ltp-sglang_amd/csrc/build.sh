#!/bin/bash
# Build the C-ABI shared library for gfx950 (hipcc cross-compiles without a GPU).
# Output: ltp-sglang_amd/lib/libsgl_mi355.so (git-ignored, travels with gpurun).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT" "$HERE/.obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -I$HERE"
# a flag change invalidates every object
if [ "$(cat "$HERE/.obj/flags.txt" 2>/dev/null)" != "$FLAGS" ]; then rm -f "$HERE"/.obj/*.o; echo "$FLAGS" > "$HERE/.obj/flags.txt"; fi
pids=()
objs=()
for src in "$HERE"/*.hip; do
  obj="$HERE/.obj/$(basename "${src%.hip}").o"
  objs+=("$obj")
  stale=0
  for h in "$HERE"/*.h; do [ "$h" -nt "$obj" ] && stale=1; done
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ "$stale" = 1 ]; then
    # extend_attention_phased.hip: no SLP vectorisation -- packed f32 adds / multiplies beside MFMAs cost more vector-port time than the
    # scalar forms (MI355X_MICROARCH.md, "price of one filler beside MFMAs"), and they drag register moves along
    EXTRA=""; [ "$(basename "$src")" = extend_attention_phased.hip ] && EXTRA="-fno-slp-vectorize"
    $HIPCC $FLAGS $EXTRA -c "$src" -o "$obj" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libsgl_mi355.so" "${objs[@]}"
echo "built $OUT/libsgl_mi355.so"
