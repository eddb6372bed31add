# Build libpaoship.so (gfx950) and the CPU-side helpers.  `python -c "import __graft_entry__ as g; g.build()"`
# runs the same commands.
HIPCC ?= hipcc
ARCH ?= gfx950
# EXTRA: experiment flags for every translation unit of the library, e.g. make EXTRA=-DPAOS_BR=8
HIPFLAGS = -O3 --offload-arch=$(ARCH) -ffp-contract=off -fPIC -std=c++17 -Wall -Wno-unused-function $(EXTRA)
CSRC = paos_amd/csrc
LIB = paos_amd/libpaoship.so

all: $(LIB)

# The library is one source file compiled as six translation units (make -j6: ~1.5 min instead of 4):
# part 0 = everything but the frugal pass-kernel families, parts 1..5 = one (type, N) family each.
DEPS = $(CSRC)/paos_hip.hip $(CSRC)/fft_core.h $(CSRC)/fft_kernels.h $(CSRC)/frugal_pass.h $(CSRC)/pointwise.h include/paos_hip.h
PARTS = 0 1 2 3 4 5
OBJS = $(foreach k,$(PARTS),build/obj/part$(k).o) build/obj/comm.o build/obj/plan.o build/obj/srchash.o

# What the library was built from: sha256 over its sources in this order (paos_source_hash(); __graft_entry__.build() compares
# it with the tree and rebuilds on a mismatch -- a prebuilt .so that travelled with the tree cannot silently be stale)
HASHED = $(CSRC)/paos_hip.hip $(CSRC)/fft_core.h $(CSRC)/fft_kernels.h $(CSRC)/frugal_pass.h $(CSRC)/pointwise.h \
         $(CSRC)/paos_comm.cpp $(CSRC)/paos_plan.cpp include/paos_hip.h include/paos_comm.h include/paos_plan.h
build/obj/srchash.o: $(HASHED)
	mkdir -p build/obj
	printf 'extern "C" const char* paos_source_hash(void) { return "%s"; }\n' "$$(cat $(HASHED) | sha256sum | cut -c1-32)" > build/obj/srchash.cpp
	g++ -O2 -fPIC -c build/obj/srchash.cpp -o $@

# the scalar half of the propagation loop for a batch (include/paos_plan.h): plain C++, the
# reference's operation order, no FMA contraction
build/obj/plan.o: $(CSRC)/paos_plan.cpp include/paos_plan.h
	mkdir -p build/obj
	g++ -O2 -fPIC -std=c++17 -Wall -ffp-contract=off -c $(CSRC)/paos_plan.cpp -o $@

# the multi-GPU fan-out (include/paos_comm.h): host code only, RCCL is dlopen'ed at run time
build/obj/comm.o: $(CSRC)/paos_comm.cpp include/paos_comm.h include/paos_hip.h
	mkdir -p build/obj
	$(HIPCC) -O2 -fPIC -std=c++17 -Wall -c $(CSRC)/paos_comm.cpp -o $@

build/obj/part%.o: $(DEPS)
	mkdir -p build/obj
	$(HIPCC) $(HIPFLAGS) -DPAOS_PART=$* -I$(CSRC) -c $(CSRC)/paos_hip.hip -o $@

$(LIB): $(OBJS)
	$(HIPCC) -shared -fPIC $(OBJS) -ldl -o $(LIB)

build/fftbench: tools/fftbench.hip $(CSRC)/fft_core.h $(CSRC)/fft_kernels.h $(CSRC)/frugal_pass.h
	mkdir -p build
	$(HIPCC) -O3 --offload-arch=$(ARCH) -ffp-contract=off -I$(CSRC) tools/fftbench.hip -o build/fftbench

# every pass-kernel shape should fit its register budget without scratch: rebuild with the compiler's resource remarks
# and list the shapes that spill (a change that costs a shape its allocation shows here, not only in the bench)
spillcheck:
	touch $(CSRC)/paos_hip.hip
	$(MAKE) -j6 EXTRA=-Rpass-analysis=kernel-resource-usage > build/make.log 2>&1
	python3 tools/spill_report.py build/make.log

clean:
	rm -f $(LIB) $(OBJS) build/fftbench
