#!/bin/bash
# Dynamic vector-instruction mix of the 4096^2 complex128 pass shapes (tools/fftbench.hip, PAOS_BENCH_CORE).
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/valu_round
BIN=${1:-fftbench_new}
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export PAOS_BENCH_CORE=1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT -d "$OUT/a" -o pmc --output-format csv -- "$ROOT/build/$BIN" 3 > "$OUT/fftbench_a.log" 2>&1 || exit 1
python3 "$ROOT/tools/pmc_summary.py" $(find "$OUT/a" -name "*counter_collection.csv" | head -1) > "$OUT/pmc_a.txt" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU -d "$OUT/b" -o pmc --output-format csv -- "$ROOT/build/$BIN" 3 > "$OUT/fftbench_b.log" 2>&1 || exit 1
python3 "$ROOT/tools/pmc_summary.py" $(find "$OUT/b" -name "*counter_collection.csv" | head -1) > "$OUT/pmc_b.txt" 2>&1
find "$OUT" -name "*.csv" -size +8M -delete
