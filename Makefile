# Build the HIP engine (gfx950 only) in-tree.  `make` == what __graft_entry__.build() runs.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := pclsegmentation_amd/csrc
LIB   := pclsegmentation_amd/libpclseg.so
SRCS  := $(CSRC)/pclseg_kernels.h $(CSRC)/pclseg_graph.h $(CSRC)/pclseg_api.hip include/pclseg.h
# sha256 over the sources, in the order bench.py's csrc_sha() reads them, baked into the library
# (pclseg_build_sha): bench.py compares it with the sources next to it and refuses to quote a PMC
# traffic figure when the binary that ran was built from something else
SRC_SHA := $(shell cat $(SRCS) | sha256sum | cut -c1-16)
HIPFLAGS := -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -shared -Wall -Wno-unused-function -DPCLSEG_SRC_SHA=\"$(SRC_SHA)\" $(EXTRA)

all: $(LIB)

$(LIB): $(SRCS)
	$(HIPCC) $(HIPFLAGS) -o $@ $(CSRC)/pclseg_api.hip

# debug build with in-kernel phase timestamps (PCLSEG_STAMP=<layer> PCLSEG_LIB=.../libpclseg_stamps.so)
stamps: $(SRCS)
	$(HIPCC) $(HIPFLAGS) -DPCLSEG_WITH_STAMPS -o pclsegmentation_amd/libpclseg_stamps.so $(CSRC)/pclseg_api.hip

# A/B build: the experiment switches of DESIGN.md §9/§10 (PCLSEG_GEOM, PCLSEG_DN8, ...) are read from the
# environment (PCLSEG_LIB=.../libpclseg_tuning.so); the shipped library carries only their defaults
tuning: $(SRCS)
	$(HIPCC) $(HIPFLAGS) -DPCLSEG_TUNING -o pclsegmentation_amd/libpclseg_tuning.so $(CSRC)/pclseg_api.hip

# experiment build of the round: kernel variants that have NOT yet been verified on an MI355X are compiled only
# with -DPCLSEG_R4X (A/B: PCLSEG_LIB=.../libpclseg_r4x.so); the shipped library carries the verified code paths
r4x: $(SRCS)
	$(HIPCC) $(HIPFLAGS) -DPCLSEG_R4X -o pclsegmentation_amd/libpclseg_r4x.so $(CSRC)/pclseg_api.hip

# any other A/B build: make variant NAME=r4x_noslab EXTRA="-DPCLSEG_R4X_TAIL -DPCLSEG_R4X_CAM -DPCLSEG_R4X_WIDE"
variant: $(SRCS)
	$(HIPCC) $(HIPFLAGS) -o pclsegmentation_amd/libpclseg_$(NAME).so $(CSRC)/pclseg_api.hip

clean:
	rm -f $(LIB) pclsegmentation_amd/libpclseg_stamps.so pclsegmentation_amd/libpclseg_tuning.so pclsegmentation_amd/libpclseg_r4x.so pclsegmentation_amd/libpclseg_base.so

.PHONY: all clean stamps tuning r4x variant
