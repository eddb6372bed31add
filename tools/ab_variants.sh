#!/bin/bash
# ab_variants.sh ROUNDS NAME...: bench.py (4096^2 complex128, 20 steps) with each build/variants/NAME.so in turn, ROUNDS times.
ROUNDS=$1; shift
OUT=gpurun_out/ab; mkdir -p $OUT
cp paos_amd/libpaoship.so $OUT/shipped.so
for r in $(seq $ROUNDS); do
  for v in "$@"; do
    cp build/variants/$v.so paos_amd/libpaoship.so
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-traffic > $OUT/${v}_$r.json 2>/dev/null || exit 1
    python - $OUT/${v}_$r.json $v $r <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print(f"{sys.argv[2]:10s} round {sys.argv[3]}: {d['value']:.1f} wavefronts/s  full launches {r['avg_launch_ms']*1e3:.1f} us  pruned {r['pruned']['avg_launch_ms']*1e3:.1f} us")
PY
  done
done
cp $OUT/shipped.so paos_amd/libpaoship.so; rm -f $OUT/shipped.so
