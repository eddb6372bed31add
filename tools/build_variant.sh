#!/bin/bash
# build_variant.sh NAME "-DFLAG=.. ..." : libpaoship with the 4096^2 complex128 pass family (part 3) compiled with extra
# flags -> build/variants/NAME.so.  For A/B runs of bench.py on one box (copy the variant over paos_amd/libpaoship.so
# in the gpurun command): bench.py repeats to +-0.1 %, tools/fftbench.hip only to +-1.5 %.
set -e
NAME=$1; FLAGS=$2
mkdir -p build/variants
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -Wall -Wno-unused-function $FLAGS -DPAOS_PART=3 -Ipaos_amd/csrc -c paos_amd/csrc/paos_hip.hip -o build/variants/part3_$NAME.o
/opt/rocm/bin/hipcc -shared -fPIC build/obj/part0.o build/obj/part1.o build/obj/part2.o build/variants/part3_$NAME.o build/obj/part4.o build/obj/part5.o build/obj/comm.o build/obj/plan.o -ldl -o build/variants/$NAME.so
rm -f build/variants/part3_$NAME.o
