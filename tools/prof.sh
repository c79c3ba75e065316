#!/bin/bash
# tools/prof.sh <outdir> <frames.py args...>: kernel trace of undisturbed frames on the GPU box + per-kernel breakdown.
# (run from the repository root on the GPU box)
out=$1; shift
root=$(pwd)
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$root/$out" -- python3 "$root/tools/frames.py" "$@" > "$root/$out/frames.log" 2> "$root/$out/prof.err"
rc=$?
cd "$root"
cat "$out/frames.log"
python3 tools/kstats.py $(ls $out/*/*_kernel_trace.csv | head -1) 80 | cut -c1-100
exit $rc
