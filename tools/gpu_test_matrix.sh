#!/bin/bash
# tools/gpu_test_matrix.sh <outdir>: the GPU suite (one process each) under every environment switch of the library that selects
# another code path, on the box this runs on (about a minute each: 21 switches + 3 shuffled orders need two
# gpurun calls of 20 minutes: `tools/gpu_test_matrix.sh out shuffle` runs only the shuffled orders).  A summary line per switch goes
# to <outdir>/matrix.txt.
out=$1; mkdir -p "$out"; : > "$out/matrix.txt"
run() { # label, env assignments...
  label=$1; shift
  log="$out/$(echo "$label" | tr ' =' '__').log"
  env "$@" timeout -k 10 400 python -m pytest tests -m gpu -q > "$log" 2>&1
  printf '%-34s %s\n' "$label" "$(tail -1 "$log")" >> "$out/matrix.txt"
}
if [ "$2" != "shuffle" ]; then
run "default" SPLAT_NOOP=1
run "SPLAT_RANK=ballot" SPLAT_RANK=ballot
run "SPLAT_FRAME_ORDER=sortfirst" SPLAT_FRAME_ORDER=sortfirst
run "SPLAT_TILE_SORT_DIGITS=12" SPLAT_TILE_SORT_DIGITS=12
run "SPLAT_TILE_SORT_CLASSES=1" SPLAT_TILE_SORT_CLASSES=1
run "SPLAT_TILE_SORT_SHORT=8" SPLAT_TILE_SORT_SHORT=8
run "SPLAT_TILE_SORT_SHORT=12" SPLAT_TILE_SORT_SHORT=12
run "SPLAT_TILE_SORT_SHORT=16" SPLAT_TILE_SORT_SHORT=16
run "SPLAT_COMPOSITE=pixel" SPLAT_COMPOSITE=pixel
run "SPLAT_COMPOSITE=quadrant" SPLAT_COMPOSITE=quadrant
run "SPLAT_TILE_ORDER=0" SPLAT_TILE_ORDER=0
run "SPLAT_PX_AHEAD=2" SPLAT_PX_AHEAD=2
run "SPLAT_PX_AHEAD=1" SPLAT_PX_AHEAD=1
run "SPLAT_PX_PREDICT=0" SPLAT_PX_PREDICT=0
run "SPLAT_PX_SLACK=1" SPLAT_PX_SLACK=1
run "SPLAT_BAND_COMPACT=1" SPLAT_BAND_COMPACT=1
run "SPLAT_BAND_COMPACT=0" SPLAT_BAND_COMPACT=0
run "SPLAT_BAND_RECORDS=lit" SPLAT_BAND_RECORDS=lit
run "SPLAT_RADIX_MODE=onesweep" SPLAT_RADIX_MODE=onesweep
run "SPLAT_BIN_SYNC=1" SPLAT_BIN_SYNC=1
run "SPLAT_RANK=atomic" SPLAT_RANK=atomic
fi
if [ "$2" != "switches" ]; then
run "SPLAT_TEST_SHUFFLE=1" SPLAT_TEST_SHUFFLE=1
run "SPLAT_TEST_SHUFFLE=2" SPLAT_TEST_SHUFFLE=2
run "SPLAT_TEST_SHUFFLE=3" SPLAT_TEST_SHUFFLE=3
fi
cat "$out/matrix.txt"
