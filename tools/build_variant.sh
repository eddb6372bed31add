#!/bin/bash
# build_variant.sh NAME "-DFLAG=.. ..." : libpaoship with the 4096^2 complex128 pass family (part 3) compiled with extra
# flags -> build/ab/NAME.so (travels to the GPU box; build/variants/ does not).  For A/B runs of bench.py on one box
# (tools/ab_variants.sh selects the variant through PAOS_LIB): bench.py repeats to +-0.1 %, tools/fftbench.hip only to +-1.5 %.
set -e
NAME=$1; FLAGS=$2
mkdir -p build/ab
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -Wall -Wno-unused-function $FLAGS -DPAOS_PART=3 -Ipaos_amd/csrc -c paos_amd/csrc/paos_hip.hip -o build/ab/part3_$NAME.o
/opt/rocm/bin/hipcc -shared -fPIC build/obj/part0.o build/obj/part1.o build/obj/part2.o build/ab/part3_$NAME.o build/obj/part4.o build/obj/part5.o build/obj/comm.o build/obj/plan.o build/obj/srchash.o -ldl -o build/ab/$NAME.so
rm -f build/ab/part3_$NAME.o
