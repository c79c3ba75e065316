#!/bin/bash
# tools/px_r04_ab.sh <outfile> [configs]: round 4's composite A/B on one box — round 3's kernel (splat_renderer_amd/_variants/libsplat_r3.so,
# built from commit 1344df7 with tools/build_variant.sh) against the current k_composite_px with its three changes switched
# one by one: SPLAT_PX_AHEAD=1 (builder one chunk ahead, no lane runs ahead), SPLAT_PX_PREDICT=0 (no look-ahead bound from the
# previous launch's per-tile costs), both (= the table recurrence alone).  tools/px_ab.py times the kernel alone.
out=$1; shift
cfgs=${@:-C2}
: > "$out"
for cfg in $cfgs; do
  if [ -f splat_renderer_amd/_variants/libsplat_r3.so ]; then
    SPLAT_LIB_PATH=$PWD/splat_renderer_amd/_variants/libsplat_r3.so SPLAT_COMPOSITE=pixel python3 tools/px_ab.py $cfg >> "$out" 2>&1
  fi
  echo "[default: ahead=1 with early-out, 2 without; predict=1]" >> "$out"; SPLAT_COMPOSITE=pixel python3 tools/px_ab.py $cfg >> "$out" 2>&1
  echo "[ahead=2 predict=1]" >> "$out"; SPLAT_PX_AHEAD=2 SPLAT_COMPOSITE=pixel python3 tools/px_ab.py $cfg >> "$out" 2>&1
  echo "[ahead=1 predict=1]" >> "$out"; SPLAT_PX_AHEAD=1 SPLAT_COMPOSITE=pixel python3 tools/px_ab.py $cfg >> "$out" 2>&1
  echo "[ahead=2 predict=0]" >> "$out"; SPLAT_PX_AHEAD=2 SPLAT_PX_PREDICT=0 SPLAT_COMPOSITE=pixel python3 tools/px_ab.py $cfg >> "$out" 2>&1
  echo "[ahead=1 predict=0]" >> "$out"; SPLAT_PX_AHEAD=1 SPLAT_PX_PREDICT=0 SPLAT_COMPOSITE=pixel python3 tools/px_ab.py $cfg >> "$out" 2>&1
done
cat "$out"
