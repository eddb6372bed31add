#!/bin/bash
# ab_variants.sh ROUNDS NAME...: bench.py (4096^2 complex128, 20 steps) with each build/variants/NAME.so in turn, ROUNDS times.
# The variant is selected through PAOS_LIB (paos_amd/_lib.py), the shipped library is never overwritten.
ROUNDS=$1; shift
OUT=gpurun_out/ab; mkdir -p $OUT
for r in $(seq $ROUNDS); do
  for v in "$@"; do
    PAOS_LIB=$PWD/build/variants/$v.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-traffic > $OUT/${v}_$r.json 2>$OUT/${v}_$r.err || { tail -3 $OUT/${v}_$r.err; exit 1; }
    python - $OUT/${v}_$r.json $v $r <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print(f"{sys.argv[2]:10s} round {sys.argv[3]}: {d['value']:.1f} wavefronts/s  full launches {r['avg_launch_ms']*1e3:.1f} us  pruned {r['pruned']['avg_launch_ms']*1e3:.1f} us")
PY
  done
done
