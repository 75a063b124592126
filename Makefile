# Build the HIP engine (gfx950 only) in-tree.  `make` == what __graft_entry__.build() runs.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := pclsegmentation_amd/csrc
LIB   := pclsegmentation_amd/libpclseg.so
SRCS  := $(CSRC)/pclseg_kernels.h $(CSRC)/pclseg_graph.h $(CSRC)/pclseg_api.hip include/pclseg.h
# sha256 over the sources (in the order bench.py's csrc_sha() reads them) FOLLOWED BY the build's -D switches, baked into
# the library (pclseg_build_sha): bench.py compares it with the sources next to it and refuses to quote a PMC traffic
# figure when the binary that ran was built from something else.  The shipped build has no switches, so its hash is the
# hash of the sources alone; every stamps / tuning / variant build hashes differently and is never mistaken for it.
sha = $(shell (cat $(SRCS); printf '%s' '$(strip $(1))') | sha256sum | cut -c1-16)
HIPFLAGS := -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -shared -Wall -Wno-unused-function
# $(call hipbuild,<output>,<-D switches>)
hipbuild = $(HIPCC) $(HIPFLAGS) $(2) -DPCLSEG_SRC_SHA=\"$(call sha,$(2))\" -o $(1) $(CSRC)/pclseg_api.hip

all: $(LIB)

$(LIB): $(SRCS)
	$(call hipbuild,$@,$(EXTRA))

# the candidate kernel variants of pclseg_kernels.h (never run on an MI355X; NOT in the shipped library): the
# hardware A/B of scripts/gpu_all.sh (step 4) loads this one through PCLSEG_DEBUG=1 PCLSEG_LIB=build/libpclseg_cand.so
candidates: $(SRCS)
	@mkdir -p build
	$(call hipbuild,build/libpclseg_cand.so,-DPCLSEG_CAND $(EXTRA))

# debug build with in-kernel phase timestamps (PCLSEG_STAMP=<layer>; loaded with PCLSEG_DEBUG=1 PCLSEG_LIB=build/libpclseg_stamps.so)
stamps: $(SRCS)
	@mkdir -p build
	$(call hipbuild,build/libpclseg_stamps.so,-DPCLSEG_WITH_STAMPS $(EXTRA))

# A/B build: the experiment switches of DESIGN.md §9/§10 (PCLSEG_GEOM, PCLSEG_DN8, ...) are read from the
# environment; the shipped library carries only their defaults
tuning: $(SRCS)
	@mkdir -p build
	$(call hipbuild,build/libpclseg_tuning.so,-DPCLSEG_TUNING $(EXTRA))

# any other A/B build: make variant NAME=foo EXTRA="-DPCLSEG_..."  ->  build/libpclseg_foo.so
variant: $(SRCS)
	@mkdir -p build
	$(call hipbuild,build/libpclseg_$(NAME).so,$(EXTRA))

# after a host-only source edit: re-key round 3's measured PMC traffic to the new source hash (refused by the script if any
# shipped device kernel differs from the measured build; tests/test_host.py checks file and sources agree)
carry: all
	python3 scripts/carry_traffic.py profiles/r03_traffic.json ad2e081 profiles/r06_traffic.json $(wildcard profiles/r03_*_kernel_stats.csv)

clean:
	rm -f $(LIB) pclsegmentation_amd/libpclseg_*.so
	rm -rf build sim/_build

.PHONY: all clean stamps tuning variant candidates carry
