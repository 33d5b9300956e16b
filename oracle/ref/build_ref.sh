#!/bin/bash
# Builds oracle/_ref/libba_ref.so from the reference's own functor source, where it lies.
# The cut-out text lives only in a temporary directory for the duration of the compile; only
# the shared object lands under oracle/_ref/ (git-ignored, travels to the GPU box).
set -euo pipefail
REF=${BA_REFERENCE_ROOT:-/root/reference}
SRC="$REF/src/main.cpp"
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../_ref"
[ -f "$SRC" ] || { echo "build_ref: $SRC not present, skipping" >&2; exit 0; }
TMP="$(mktemp -d)"
trap 'rm -rf "$TMP"' EXIT
sed -n '55p'         "$SRC" > "$TMP/slice_apscale.inc"
sed -n '1057,1085p'  "$SRC" > "$TMP/slice_constrainPoint.inc"
sed -n '1315,1401p'  "$SRC" > "$TMP/slice_collinearity.inc"
sed -n '2033,2120p'  "$SRC" > "$TMP/slice_collinearityMLS.inc"
grep -q 'define APSCALE'                         "$TMP/slice_apscale.inc"
grep -q 'struct constrainPoint'                  "$TMP/slice_constrainPoint.inc"
grep -q 'struct collinearity {'                  "$TMP/slice_collinearity.inc"
grep -q 'struct collinearityMachineLearnedSimple' "$TMP/slice_collinearityMLS.inc"
mkdir -p "$OUT"
# -ffp-contract=off: keep the reference's expression order, no fused multiply-add
g++ -O2 -std=c++14 -ffp-contract=off -fPIC -shared -I"$TMP" "$HERE/ref_wrap.cpp" -o "$OUT/libba_ref.so"
echo "built $OUT/libba_ref.so"
