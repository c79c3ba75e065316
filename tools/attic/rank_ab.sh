#!/bin/bash
# tools/rank_ab.sh <outfile>: frame time of C0..C3 with the two rankings of the list-producing kernels, alternating
# twice on ONE box (boxes of the pool differ by 3-5 %): SPLAT_RANK=ballot (lane order by construction: ballots +
# mbcnt) against SPLAT_RANK=atomic (returning LDS atomics, stable only if colliding lanes complete in lane order).
out=$1
: > "$out"
for c in C2 C0 C1 C3; do
  for rep in 1 2; do
    for mode in atomic ballot; do
      echo -n "SPLAT_RANK=$mode " >> "$out"
      SPLAT_RANK=$mode python3 tools/frames.py $c 60 >> "$out" 2>&1 || echo "failed" >> "$out"
    done
  done
done
