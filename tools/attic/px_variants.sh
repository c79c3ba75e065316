#!/bin/bash
# tools/px_variants.sh <outfile> [config]: tools/px_ab.py for round 2's kernel, the default build of k_composite_px and
# every build under splat_renderer_amd/_variants/, on one box.
out=$1; cfg=${2:-C2}
: > "$out"
SPLAT_COMPOSITE=quadrant python3 tools/px_ab.py $cfg >> "$out" 2>&1
SPLAT_COMPOSITE=pixel python3 tools/px_ab.py $cfg >> "$out" 2>&1
for lib in splat_renderer_amd/_variants/libsplat_*.so; do
  [ -f "$lib" ] || continue
  SPLAT_LIB_PATH=$PWD/$lib SPLAT_COMPOSITE=pixel python3 tools/px_ab.py $cfg >> "$out" 2>&1
done
cat "$out"
