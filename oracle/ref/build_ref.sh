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
sed -n '2133,2214p'  "$SRC" > "$TMP/slice_stereographicMLS.inc"
sed -n '2226,2301p'  "$SRC" > "$TMP/slice_equidistantMLS.inc"
sed -n '2312,2387p'  "$SRC" > "$TMP/slice_equisolidMLS.inc"
sed -n '2398,2473p'  "$SRC" > "$TMP/slice_orthographicMLS.inc"
sed -n '2484,2562p'  "$SRC" > "$TMP/slice_fisheyeStereoMLS.inc"
sed -n '1412,1490p'  "$SRC" > "$TMP/slice_orthographic.inc"
sed -n '2700,2790p'  "$SRC" > "$TMP/slice_omniMLS.inc"
sed -n '39p'         "$SRC" > "$TMP/slice_pi.inc"
sed -n '791,823p'    "$SRC" > "$TMP/slice_stats1.inc"
sed -n '884,1019p'   "$SRC" > "$TMP/slice_stats2.inc"
grep -q 'define APSCALE'                         "$TMP/slice_apscale.inc"
grep -q 'struct constrainPoint'                  "$TMP/slice_constrainPoint.inc"
grep -q 'struct collinearity {'                  "$TMP/slice_collinearity.inc"
grep -q 'struct collinearityMachineLearnedSimple' "$TMP/slice_collinearityMLS.inc"
grep -q 'struct collinearityStereographicMachineLearnedSimple' "$TMP/slice_stereographicMLS.inc"
grep -q 'struct fisheyeEquidistantMachineLearnedSimple'        "$TMP/slice_equidistantMLS.inc"
grep -q 'struct fisheyeEquisolidAngleMachineLearnedSimple'     "$TMP/slice_equisolidMLS.inc"
grep -q 'struct fisheyeOrthographicMachineLearnedSimple'       "$TMP/slice_orthographicMLS.inc"
grep -q 'struct fisheyeStereographicMachineLearnedSimple'      "$TMP/slice_fisheyeStereoMLS.inc"
grep -q 'struct collinearityOrthographic {'                    "$TMP/slice_orthographic.inc"
grep -q 'struct omniCollinearityMachineLearnedSimple'          "$TMP/slice_omniMLS.inc"
grep -q 'define PI'                                            "$TMP/slice_pi.inc"
grep -q 'double calcMedian'                                    "$TMP/slice_stats1.inc"
grep -q 'double calculateBIC'                                  "$TMP/slice_stats2.inc"
mkdir -p "$OUT"
# -ffp-contract=off: keep the reference's expression order, no fused multiply-add
g++ -O2 -std=c++14 -ffp-contract=off -fPIC -shared -I"$TMP" "$HERE/ref_wrap.cpp" -o "$OUT/libba_ref.so"
echo "built $OUT/libba_ref.so"
