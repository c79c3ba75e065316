#!/bin/bash
# tools/gpu_test_matrix.sh <outdir>: the GPU suite (one process each) under the environment switches of the library that select
# another code path, on the box this runs on — ONCE per round, at its end (about a minute per run, eleven
# runs).  Independent switches share a run; the runs that do not touch the sort's ranking leave out the radix-sort and SDF cases
# (they take no switch).  A summary line per run goes to <outdir>/matrix.txt.
out=$1; mkdir -p "$out"; : > "$out/matrix.txt"
run() { # label, pytest -k expression ("" = everything), env assignments...
  label=$1; sel=$2; shift 2
  log="$out/$(echo "$label" | tr ' =' '__').log"
  if [ -n "$sel" ]; then env "$@" timeout -k 10 400 python -m pytest tests -m gpu -q -k "$sel" > "$log" 2>&1
  else env "$@" timeout -k 10 400 python -m pytest tests -m gpu -q > "$log" 2>&1; fi
  printf '%-52s %s\n' "$label" "$(tail -1 "$log")" >> "$out/matrix.txt"
}
rest="not radix_sort_stable and not sdf"
run "default" "" SPLAT_NOOP=1
run "SPLAT_RANK=ballot" "" SPLAT_RANK=ballot
run "SPLAT_RANK=atomic" "" SPLAT_RANK=atomic
run "SPLAT_FRAME_ORDER=sortfirst" "$rest" SPLAT_FRAME_ORDER=sortfirst
run "SPLAT_COMPOSITE=pixel" "$rest" SPLAT_COMPOSITE=pixel
run "SPLAT_COMPOSITE=quadrant SPLAT_BIN_SYNC=1" "$rest" SPLAT_COMPOSITE=quadrant SPLAT_BIN_SYNC=1
run "SPLAT_TILE_ORDER=0 SPLAT_PX_PREDICT=0" "$rest" SPLAT_TILE_ORDER=0 SPLAT_PX_PREDICT=0
run "SPLAT_PX_AHEAD=2 SPLAT_BAND_COMPACT=1" "$rest" SPLAT_PX_AHEAD=2 SPLAT_BAND_COMPACT=1
run "SPLAT_PX_AHEAD=1 SPLAT_TILE_SORT_SHORT=16" "$rest" SPLAT_PX_AHEAD=1 SPLAT_TILE_SORT_SHORT=16
run "SPLAT_BAND_COMPACT=0 SPLAT_TILE_SORT_SHORT=8" "$rest" SPLAT_BAND_COMPACT=0 SPLAT_TILE_SORT_SHORT=8
run "SPLAT_TEST_SHUFFLE=1" "" SPLAT_TEST_SHUFFLE=1
cat "$out/matrix.txt"
