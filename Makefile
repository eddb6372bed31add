# Build libpaoship.so (gfx950) and the CPU-side helpers.  `python -c "import __graft_entry__ as g; g.build()"`
# runs the same commands.
HIPCC ?= hipcc
ARCH ?= gfx950
HIPFLAGS = -O3 --offload-arch=$(ARCH) -ffp-contract=off -fPIC -std=c++17 -Wall -Wno-unused-function
CSRC = paos_amd/csrc
LIB = paos_amd/libpaoship.so

all: $(LIB)

$(LIB): $(CSRC)/paos_hip.hip $(CSRC)/fft_core.h $(CSRC)/fft_kernels.h $(CSRC)/frugal_pass.h $(CSRC)/pointwise.h include/paos_hip.h
	$(HIPCC) $(HIPFLAGS) -shared -I$(CSRC) $(CSRC)/paos_hip.hip -o $(LIB)

build/fftbench: tools/fftbench.hip $(CSRC)/fft_core.h $(CSRC)/fft_kernels.h $(CSRC)/frugal_pass.h
	mkdir -p build
	$(HIPCC) -O3 --offload-arch=$(ARCH) -ffp-contract=off -I$(CSRC) tools/fftbench.hip -o build/fftbench

clean:
	rm -f $(LIB) build/fftbench
