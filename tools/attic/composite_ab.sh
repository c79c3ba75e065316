#!/bin/bash
# tools/composite_ab.sh <outdir>: the two composite kernels on one box — k_composite_px (default) against round 2's
# k_composite (SPLAT_COMPOSITE=quadrant): bench line of C2 (lit records: the timed frame's composite and the
# early-out-off figure), and tools/composite_bench.py (ProjectedSplat records, both blend modes).
out=$1
mkdir -p "$out"
for c in ${CONFIGS:-C2}; do
  for k in pixel quadrant; do
    SPLAT_COMPOSITE=$k python3 bench.py --config $c --steps 40 --no-cpu-baseline > "$out/bench_${c}_$k.json" 2> "$out/bench_${c}_$k.err" || echo "bench $c $k failed"
    python3 - "$out/bench_${c}_$k.json" $c $k <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]; x = d.get("extra", {}).get("composite_early_out_off", {})
print(f"{sys.argv[2]} {sys.argv[3]:9s} frame {d['ms_per_step']:.4f} ms  composite {r['avg_launch_ms']*1e3:7.1f} us frac {r['frac']:.3f} "
      f"consumed {r['pairs_consumed']} staged {r['pairs_staged']}  early-out-off {x.get('avg_launch_ms', 0)*1e3:7.1f} us frac {x.get('frac', 0):.3f}  parity {d.get('parity_vs_cpu_frame')}")
PY
  done
done
for k in pixel quadrant; do
  echo "composite_bench $k"; SPLAT_COMPOSITE=$k python3 tools/composite_bench.py ${CONFIGS:-C2} 2>&1 | tail -5
done
