#!/bin/bash
# LDS / instruction-mix counters of the 4096^2 complex128 pass shapes (tools/fftbench.hip, PAOS_BENCH_CORE) for three
# builds: shipped, stage-2 twiddles multiplied up instead of read from the circle table, exchanges without LDS traffic.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/lds_round
rm -rf "$OUT" && mkdir -p "$OUT"
cd "$ROOT"
./build/ldsprobe > "$OUT/ldsprobe.txt" 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$OUT/counters_avail.txt" 2>&1
export PAOS_BENCH_CORE=1
for v in v0 notab nolds; do
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d "$OUT/$v" -o pmc --output-format csv -- "$ROOT/build/fftbench_$v" 3 > "$OUT/fftbench_$v.log" 2>&1 || exit 1
  python3 "$ROOT/tools/pmc_summary.py" $(find "$OUT/$v" -name "*counter_collection.csv" | head -1) > "$OUT/pmc_$v.txt" 2>&1
done
find "$OUT" -name "*.csv" -size +8M -delete
cat "$OUT/ldsprobe.txt"
