#!/bin/bash
# asan_host.sh: the host-only translation units of libpaoship (paos_plan.cpp: the batch planner, paos_comm.cpp: the multi-rank
# transport) rebuilt with AddressSanitizer + UBSan, linked with the ordinary kernel objects into build/asan/libpaoship_asan.so,
# and the CPU test suite run on it (PAOS_LIB selects the variant; the worker processes of the multi-rank tests inherit it).
# GPU code cannot be sanitized on this pool; these two files are where the library does its own memory management on the host.
set -e
cd "$(dirname "$0")/.."
make -s >/dev/null
mkdir -p build/asan
SAN="-O1 -g -fPIC -std=c++17 -Wall -fsanitize=address,undefined -fno-omit-frame-pointer"
g++ $SAN -ffp-contract=off -c paos_amd/csrc/paos_plan.cpp -o build/asan/plan.o
g++ $SAN -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -c paos_amd/csrc/paos_comm.cpp -o build/asan/comm.o
GCCLIB=$(dirname "$(gcc -print-file-name=libasan.so)")
/opt/rocm/bin/hipcc -shared -fPIC build/obj/part[0-5].o build/asan/comm.o build/asan/plan.o build/obj/srchash.o -ldl -L"$GCCLIB" -lasan -lubsan -o build/asan/libpaoship_asan.so
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 LD_PRELOAD="$GCCLIB/libasan.so" \
  PAOS_LIB="$PWD/build/asan/libpaoship_asan.so" python -m pytest tests -x -q -m "not gpu" "$@"
