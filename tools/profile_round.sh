#!/bin/bash
# Round profile on the GPU box (run via gpurun): per-kernel times and HBM traffic of the default bench workload,
# the pass micro-benchmark, the workgroup timeline, parity and BASELINE-config reports.  Counters are collected in
# their own passes with --kernel-trace only (MI355X_MICROARCH.md, HBM section).  Results under gpurun_out/profile_round/;
# copy the *.txt / *.json into profiles/rNN_* to keep them.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profile_round
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-traffic"
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o stats --output-format csv -- $BENCH > "$OUT/bench_under_rocprof.log" 2>&1
python3 "$ROOT/tools/kernel_stats_summary.py" "$OUT/stats" "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-traffic (4096^2 c128, batch 32; 6 chain steps of 24 passes + ptp probe + copy yardstick)" > "$OUT/kernel_stats.txt"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" -o fetch --output-format csv -- python3 "$ROOT/bench.py" --steps 2 --warmup 0 --no-cpu-baseline --no-extras --no-traffic > "$OUT/bench_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" -o write --output-format csv -- python3 "$ROOT/bench.py" --steps 2 --warmup 0 --no-cpu-baseline --no-extras --no-traffic > "$OUT/bench_write.log" 2>&1
python3 "$ROOT/tools/pmc_summary.py" $(find "$OUT/fetch" -name "*counter_collection.csv" | head -1) > "$OUT/pmc_fetch.txt"
python3 "$ROOT/tools/pmc_summary.py" $(find "$OUT/write" -name "*counter_collection.csv" | head -1) > "$OUT/pmc_write.txt"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT -d "$OUT/sq" -o sq --output-format csv -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-traffic > "$OUT/bench_sq.log" 2>&1
python3 "$ROOT/tools/pmc_summary.py" $(find "$OUT/sq" -name "*counter_collection.csv" | head -1) > "$OUT/pmc_sq.txt" 2>&1
# keep the merge-back small
find "$OUT" -name "*.csv" -size +8M -delete
cd "$ROOT"
./build/fftbench 10 > "$OUT/fftbench.txt" 2>&1
[ -x ./build/timeline ] && ./build/timeline > "$OUT/timeline.txt" 2>&1
# (round 5: stdout of bench.py is the <= 4 KB contract line; everything else goes to the --detail record)
python3 bench.py --detail "$OUT/bench_default.json" > "$OUT/bench_default_line.json" 2> "$OUT/bench_default.err"
python3 bench.py --precision fp32 --steps 10 --warmup 2 --no-cpu-baseline --no-traffic --no-extras --detail "$OUT/bench_fp32.json" > "$OUT/bench_fp32_line.json" 2>&1
PAOS_NO_PRUNE=1 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-traffic --detail "$OUT/bench_noprune.json" > "$OUT/bench_noprune_line.json" 2>&1
if [ -z "$PAOS_PROFILE_SHORT" ]; then
python3 tests/reports/parity_report.py --sizes 1024 2048 > "$OUT/parity_gpu_vs_oracle.txt" 2>&1
python3 tests/reports/run_configs.py > "$OUT/baseline_configs.txt" 2>&1
fi
head -14 "$OUT/kernel_stats.txt"
python3 tools/bench_line.py "$OUT/bench_default.json" "$OUT/bench_fp32.json" "$OUT/bench_noprune.json"
