#!/bin/bash
# tools/variant_ab.sh <out file> <config> <variant> [<variant> ...]: bench.py's frame and stage times for builds of the library made by
# tools/build_variant.sh ("-" = the tree's own build), one after the other on the box this runs on, twice round.
out=$1; cfg=$2; shift 2
mkdir -p "$(dirname "$out")"; : > "$out"
for round in 1 2; do
  for v in "$@"; do
    lib=""; [ "$v" != "-" ] && lib=splat_renderer_amd/_variants/libsplat_$v.so
    SPLAT_LIB_PATH=$lib timeout -k 10 200 python bench.py --config $cfg --no-cpu-baseline --no-parity --steps 40 > /tmp/vab.json 2> /tmp/vab.err || { echo "$v FAILED" >> "$out"; tail -3 /tmp/vab.err >> "$out"; exit 1; }
    python - "$v" >> "$out" <<'PY'
import json, sys
d = json.load(open('/tmp/vab.json'))
s = d['stage_ms']
print('%-10s frame %.4f ms | project %.4f scatter %.4f second pass %.4f tile sort %.4f composite %.4f | order faults %s' % (
    sys.argv[1], d['ms_per_step'], s['project'], s['bin_scatter'], s['bin_second_pass'], s['bin_tile_sort'], s['composite'], d['config']['ranking']['orderFaults']))
PY
  done
done
cat "$out"
