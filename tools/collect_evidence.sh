#!/bin/bash
# tools/collect_evidence.sh <outdir>: everything profiles/ is built from, in ONE call on the GPU box, ONCE per round (run from
# the repository root): bench lines of all four configurations, the N > 1 path with one rank (--band-path under torchrun) at C2
# and C3, rocprofv3 kernel trace + stats of the default bench command AND of C3 (BASELINE's rocprof-roofline configuration),
# FETCH_SIZE / WRITE_SIZE and SQ / LDS / GRBM counters in separate --pmc passes (program directly after `--`), virtual-rank
# times, layout A/B, frames in flight.
out=$1
root=$(pwd)
mkdir -p "$out"
for c in C2 C0 C1 C3; do
  python3 bench.py --config $c --steps 40 > "$out/bench_$c.json" 2> "$out/bench_$c.err" || echo "bench $c failed"
  echo "bench $c done"
done
for c in C2 C3; do
  python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 400)) \
    bench.py --gpus 1 --band-path --config $c --steps 40 --warmup 5 --collective abi --exchange allgather > "$out/bench_band_path_$c.json" 2> "$out/bench_band_path_$c.err" || echo "band path $c failed"
done
echo "band path done"
python3 bench.py --records projected --steps 40 --no-cpu-baseline > "$out/bench_C2_projected_records.json" 2>/dev/null
python3 bench.py --layout planes --steps 40 --no-cpu-baseline > "$out/bench_C2_planes_prelit.json" 2>/dev/null
python3 bench.py --footprint disc --steps 40 > "$out/bench_C2_disc.json" 2>/dev/null
echo "bench variants done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$root/$out/trace" -- python3 "$root/bench.py" --no-cpu-baseline --no-parity --no-extras --steps 100 > "$root/$out/bench_under_trace.json" 2> "$root/$out/trace.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$root/$out/trace_C3" -- python3 "$root/bench.py" --config C3 --no-cpu-baseline --no-parity --no-extras --steps 60 > "$root/$out/bench_C3_under_trace.json" 2> "$root/$out/trace_C3.err"
echo "trace done"
for pmc in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE; do
  rocprofv3 --pmc $pmc --output-format csv -d "$root/$out/pmc_$pmc" -- python3 "$root/bench.py" --no-cpu-baseline --no-parity --steps 5 > /dev/null 2> "$root/$out/pmc_$pmc.err"
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d "$root/$out/pmc_SQ" -- python3 "$root/bench.py" --no-cpu-baseline --no-parity --steps 5 > /dev/null 2> "$root/$out/pmc_SQ.err"
# the LDS side (which bound holds): instructions, array cycles, bank-conflict cycles, issue stalls
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES --output-format csv -d "$root/$out/pmc_LDS" -- python3 "$root/bench.py" --no-cpu-baseline --no-parity --steps 5 > /dev/null 2> "$root/$out/pmc_LDS.err"
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d "$root/$out/pmc_C3_GRBM_GUI_ACTIVE" -- python3 "$root/bench.py" --config C3 --no-cpu-baseline --no-parity --no-extras --steps 5 > /dev/null 2> "$root/$out/pmc_C3_GRBM.err"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d "$root/$out/pmc_C3_SQ" -- python3 "$root/bench.py" --config C3 --no-cpu-baseline --no-parity --no-extras --steps 5 > /dev/null 2> "$root/$out/pmc_C3_SQ.err"
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES --output-format csv -d "$root/$out/pmc_C3_LDS" -- python3 "$root/bench.py" --config C3 --no-cpu-baseline --no-parity --no-extras --steps 5 > /dev/null 2> "$root/$out/pmc_C3_LDS.err"
for c in C0 C1 C3; do
  for pmc in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $pmc --output-format csv -d "$root/$out/pmc_${c}_$pmc" -- python3 "$root/bench.py" --config $c --no-cpu-baseline --no-parity --steps 5 > /dev/null 2> "$root/$out/pmc_${c}_$pmc.err"
  done
done
echo "pmc done"
cd "$root"
python3 tools/kstats.py $(ls $out/trace/*/*_kernel_trace.csv | head -1) 120 > "$out/kernel_breakdown.txt"
cp $(ls $out/trace/*/*_kernel_stats.csv | head -1) "$out/kernel_stats.csv"
python3 tools/pmc_merge.py $(ls $out/pmc_FETCH_SIZE/*/*_counter_collection.csv | head -1) $(ls $out/pmc_WRITE_SIZE/*/*_counter_collection.csv | head -1) > "$out/pmc_fetch_write.csv"
python3 tools/pmc_avg.py --valu $(ls $out/pmc_SQ/*/*_counter_collection.csv | head -1) $(ls $out/pmc_GRBM_GUI_ACTIVE/*/*_counter_collection.csv | head -1) > "$out/sq_counters.csv"
python3 tools/pmc_avg.py $(ls $out/pmc_LDS/*/*_counter_collection.csv | head -1) $(ls $out/pmc_GRBM_GUI_ACTIVE/*/*_counter_collection.csv | head -1) > "$out/lds_counters.csv"
python3 tools/kstats.py $(ls $out/trace_C3/*/*_kernel_trace.csv | head -1) 70 > "$out/kernel_breakdown_C3.txt"
cp $(ls $out/trace_C3/*/*_kernel_stats.csv | head -1) "$out/kernel_stats_C3.csv"
python3 tools/pmc_avg.py --valu $(ls $out/pmc_C3_SQ/*/*_counter_collection.csv | head -1) $(ls $out/pmc_C3_GRBM_GUI_ACTIVE/*/*_counter_collection.csv | head -1) > "$out/sq_counters_C3.csv"
python3 tools/pmc_avg.py $(ls $out/pmc_C3_LDS/*/*_counter_collection.csv | head -1) $(ls $out/pmc_C3_GRBM_GUI_ACTIVE/*/*_counter_collection.csv | head -1) > "$out/lds_counters_C3.csv"
for c in C0 C1 C3; do
  python3 tools/pmc_merge.py $(ls $out/pmc_${c}_FETCH_SIZE/*/*_counter_collection.csv | head -1) $(ls $out/pmc_${c}_WRITE_SIZE/*/*_counter_collection.csv | head -1) > "$out/pmc_fetch_write_$c.csv"
done
BAND_LAYOUT=interleaved python3 tools/band_bench.py C2 1 2 4 8 > "$out/virtual_rank_times.txt" 2>&1
BAND_LAYOUT=interleaved python3 tools/band_bench.py C3 1 8 >> "$out/virtual_rank_times.txt" 2>&1
python3 tools/replicated_bench.py C2 1 2 4 8 > "$out/exchange_free_rank_times.txt" 2>&1
python3 tools/layout_ab.py C2 100 > "$out/layout_ab_C2.txt" 2>&1
python3 tools/two_in_flight.py C2 > "$out/frames_in_flight.txt" 2>&1
python3 tools/two_in_flight.py C1 >> "$out/frames_in_flight.txt" 2>&1
rm -rf "$out"/pmc_*/ "$out/trace" "$out/trace_C3"
echo "evidence collected in $out"
