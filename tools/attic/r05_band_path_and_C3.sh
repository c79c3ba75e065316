#!/bin/bash
# tools/attic/r05_band_path_and_C3.sh <outdir>: VERDICT r4 items 1 and 2 in one call on the GPU box (run from the repository root):
# bench.py's N > 1 path with one rank under torch.distributed.run (C2, C3; all-gather cut and --exchange auto), then C3's
# kernel trace and counters (SQ, LDS, FETCH/WRITE) with the program directly after `--`.
out=$1
root=$(pwd)
mkdir -p "$out"
run_band() {  # <config> <tag> <extra args...>
  c=$1; tag=$2; shift 2
  python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 400)) \
    bench.py --gpus 1 --band-path --config $c --steps 40 --warmup 5 "$@" > "$out/bench_band_path_${c}_$tag.json" 2> "$out/bench_band_path_${c}_$tag.err" \
    || { echo "band path $c $tag FAILED"; tail -20 "$out/bench_band_path_${c}_$tag.err"; return 1; }
  echo "band path $c $tag done"
}
run_band C2 abi_allgather --collective abi --exchange allgather &&
run_band C2 abi_auto --collective abi --exchange auto &&
run_band C3 abi_allgather --collective abi --exchange allgather &&
run_band C3 abi_auto --collective abi --exchange auto &&
run_band C2 torch_allgather --collective torch --exchange allgather || exit 1
for c in C2 C3; do
  python3 bench.py --config $c --steps 40 --no-cpu-baseline > "$out/bench_$c.json" 2> "$out/bench_$c.err" || { echo "bench $c failed"; exit 1; }
  echo "bench $c done"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$root/$out/trace_C3" -- python3 "$root/bench.py" --config C3 --no-cpu-baseline --no-parity --no-extras --steps 60 > "$root/$out/bench_C3_under_trace.json" 2> "$root/$out/trace_C3.err" || { echo "trace failed"; exit 1; }
echo "trace done"
for pmc in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE; do
  rocprofv3 --pmc $pmc --output-format csv -d "$root/$out/pmc_C3_$pmc" -- python3 "$root/bench.py" --config C3 --no-cpu-baseline --no-parity --no-extras --steps 5 > /dev/null 2> "$root/$out/pmc_C3_$pmc.err" || { echo "pmc $pmc failed"; exit 1; }
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d "$root/$out/pmc_C3_SQ" -- python3 "$root/bench.py" --config C3 --no-cpu-baseline --no-parity --no-extras --steps 5 > /dev/null 2> "$root/$out/pmc_C3_SQ.err" || { echo "pmc SQ failed"; exit 1; }
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES --output-format csv -d "$root/$out/pmc_C3_LDS" -- python3 "$root/bench.py" --config C3 --no-cpu-baseline --no-parity --no-extras --steps 5 > /dev/null 2> "$root/$out/pmc_C3_LDS.err" || { echo "pmc LDS failed"; exit 1; }
echo "pmc done"
cd "$root"
python3 tools/kstats.py $(ls $out/trace_C3/*/*_kernel_trace.csv | head -1) 70 > "$out/C3_kernel_breakdown.txt"
cp $(ls $out/trace_C3/*/*_kernel_stats.csv | head -1) "$out/C3_kernel_stats.csv"
python3 tools/pmc_merge.py $(ls $out/pmc_C3_FETCH_SIZE/*/*_counter_collection.csv | head -1) $(ls $out/pmc_C3_WRITE_SIZE/*/*_counter_collection.csv | head -1) > "$out/C3_pmc_fetch_write.csv"
python3 tools/pmc_avg.py --valu $(ls $out/pmc_C3_SQ/*/*_counter_collection.csv | head -1) $(ls $out/pmc_C3_GRBM_GUI_ACTIVE/*/*_counter_collection.csv | head -1) > "$out/C3_sq_counters.csv"
python3 tools/pmc_avg.py $(ls $out/pmc_C3_LDS/*/*_counter_collection.csv | head -1) $(ls $out/pmc_C3_GRBM_GUI_ACTIVE/*/*_counter_collection.csv | head -1) > "$out/C3_lds_counters.csv"
rm -rf "$out"/pmc_C3_*/ "$out/trace_C3"
echo "collected in $out"
