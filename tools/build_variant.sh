#!/bin/bash
# build_variant.sh NAME "-DFLAG=.. ..." : libpaoship with the 4096^2 complex128 pass family (part 3) compiled with extra
# flags -> build/ab/NAME.so (travels to the GPU box; build/variants/ does not).  For A/B runs of bench.py on one box
# (tools/ab_variants.sh selects the variant through PAOS_LIB): bench.py repeats to +-0.1 %, tools/fftbench.hip only to +-1.5 %.
set -e
NAME=$1; FLAGS=$2; PART=${3:-3}   # PART: 3 = double 4096 (default), 2 = double 2048, 1 = double 1024, 4 / 5 = float 2048 / 4096
mkdir -p build/ab
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -Wall -Wno-unused-function $FLAGS -DPAOS_PART=$PART -Ipaos_amd/csrc -c paos_amd/csrc/paos_hip.hip -o build/ab/part${PART}_$NAME.o
OBJS=""
for k in 0 1 2 3 4 5; do if [ $k = $PART ]; then OBJS="$OBJS build/ab/part${PART}_$NAME.o"; else OBJS="$OBJS build/obj/part$k.o"; fi; done
/opt/rocm/bin/hipcc -shared -fPIC $OBJS build/obj/comm.o build/obj/plan.o build/obj/srchash.o -ldl -o build/ab/$NAME.so
rm -f build/ab/part${PART}_$NAME.o
