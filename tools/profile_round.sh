#!/bin/bash
# Round profile on the GPU box: per-kernel times and HBM traffic of the default bench (run via gpurun).
# Counters are collected in their own passes with --kernel-trace only (MI355X_MICROARCH.md, HBM section).
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profile_round
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o stats --output-format csv -- python3 "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu-baseline > "$OUT/bench_under_rocprof.log" 2>&1
python3 "$ROOT/tools/kernel_stats_summary.py" "$OUT/stats" "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline (4096^2 c128, batch 8; 6 chain steps + ptp probe)" > "$OUT/kernel_stats.txt"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" -o fetch --output-format csv -- python3 "$ROOT/bench.py" --steps 2 --warmup 0 --no-cpu-baseline > "$OUT/bench_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" -o write --output-format csv -- python3 "$ROOT/bench.py" --steps 2 --warmup 0 --no-cpu-baseline > "$OUT/bench_write.log" 2>&1
python3 "$ROOT/tools/pmc_summary.py" $(find "$OUT/fetch" -name "*counter_collection.csv" | head -1) > "$OUT/pmc_fetch.txt"
python3 "$ROOT/tools/pmc_summary.py" $(find "$OUT/write" -name "*counter_collection.csv" | head -1) > "$OUT/pmc_write.txt"
# keep the merge-back small
find "$OUT" -name "*.csv" -size +8M -delete
head -12 "$OUT/kernel_stats.txt"
