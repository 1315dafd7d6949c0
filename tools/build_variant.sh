#!/bin/bash
# Build a VARIANT of libsgl_mi355.so with extra compiler flags (e.g. -DSGL_SKINNY_PD=1) into ltp-sglang_amd/lib/exp/<name>.so for a
# same-box A/B with tools/ab_libs.sh; the product library and its objects are not touched.
#   tools/build_variant.sh pd1 -DSGL_SKINNY_PD=1
set -euo pipefail
NAME="$1"; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC="$ROOT/ltp-sglang_amd/csrc"
OBJ="$SRC/.obj/variant_$NAME"
OUT="$ROOT/ltp-sglang_amd/lib/exp"
mkdir -p "$OBJ" "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
pids=()
for src in "$SRC"/*.hip; do
  EXTRA=""; [ "$(basename "$src")" = extend_attention_phased.hip ] && EXTRA="-fno-slp-vectorize"
  $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -I"$SRC" $EXTRA "$@" -c "$src" -o "$OBJ/$(basename "${src%.hip}").o" &
  pids+=($!)
  if [ ${#pids[@]} -ge 6 ]; then wait "${pids[0]}"; pids=("${pids[@]:1}"); fi
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/$NAME.so" "$OBJ"/*.o
echo "built $OUT/$NAME.so"
