#!/bin/bash
# quick_profile.sh: per-kernel times (rocprofv3 --kernel-trace --stats) and SQ counters of the short bench, then the
# default bench line and the parity report -- a subset of profile_round.sh for use between changes.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/quick_profile
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-traffic"
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o stats --output-format csv -- $BENCH > "$OUT/bench_under_rocprof.log" 2>&1
python3 "$ROOT/tools/kernel_stats_summary.py" "$OUT/stats" "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-traffic" > "$OUT/kernel_stats.txt"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT -d "$OUT/sq" -o sq --output-format csv -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-traffic > "$OUT/bench_sq.log" 2>&1
python3 "$ROOT/tools/pmc_summary.py" $(find "$OUT/sq" -name "*counter_collection.csv" | head -1) > "$OUT/pmc_sq.txt" 2>&1
find "$OUT" -name "*.csv" -size +8M -delete
cd "$ROOT"
python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python3 tests/reports/parity_report.py --sizes 1024 2048 > "$OUT/parity_gpu_vs_oracle.txt" 2>&1
head -20 "$OUT/kernel_stats.txt"
