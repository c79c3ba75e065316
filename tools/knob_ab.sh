#!/bin/bash
# tools/knob_ab.sh <outdir> "<configs>" "<pytest -k expression or ->" "<env settings 1>" "<env settings 2>" ...
# A/B of environment knobs on ONE box: for every setting ("" = the default) first the correctness subset (pytest -k, "-" skips
# it), then per configuration a rocprofv3 kernel trace of 40 undisturbed frames (tools/prof.sh): ms/frame and us per kernel.
out=$1; configs=$2; tests=$3; shift 3
mkdir -p "$out"
if [ "$tests" != "-" ]; then
  for v in "$@"; do
    tag=$(echo "base $v" | tr ' =/.' '____')
    env $v python3 -m pytest tests/test_gpu_stages.py -q -x -k "$tests" > "$out/test_$tag.log" 2>&1 || { echo "tests FAILED for '$v'"; tail -15 "$out/test_$tag.log"; exit 1; }
    echo "tests ok: '$v': $(tail -1 $out/test_$tag.log)"
  done
fi
for c in $configs; do
  for v in "$@"; do
    tag=$(echo "${c}_base $v" | tr ' =/.' '____')
    env $v bash tools/prof.sh "$out/$tag" $c 40 > "$out/$tag.txt" 2>&1 || { echo "prof failed: $tag"; tail -5 "$out/$tag.txt"; exit 1; }
    echo "== $c '$v'"; grep -E "ms/frame|k_|sum of kernel" "$out/$tag.txt" | grep -v "fillBuffer\|probe_lds\|scan_single"
    rm -rf "$out/$tag"
  done
done
