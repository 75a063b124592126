# Build the HIP engine (gfx950 only) in-tree.  `make` == what __graft_entry__.build() runs.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := pclsegmentation_amd/csrc
LIB   := pclsegmentation_amd/libpclseg.so
HIPFLAGS := -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -shared -Wall -Wno-unused-function

all: $(LIB)

$(LIB): $(CSRC)/pclseg_api.hip $(CSRC)/pclseg_kernels.h $(CSRC)/pclseg_graph.h include/pclseg.h
	$(HIPCC) $(HIPFLAGS) -o $@ $(CSRC)/pclseg_api.hip

# debug build with in-kernel phase timestamps (PCLSEG_STAMP=<layer> PCLSEG_LIB=.../libpclseg_stamps.so)
stamps: $(CSRC)/pclseg_api.hip $(CSRC)/pclseg_kernels.h $(CSRC)/pclseg_graph.h include/pclseg.h
	$(HIPCC) $(HIPFLAGS) -DPCLSEG_WITH_STAMPS -o pclsegmentation_amd/libpclseg_stamps.so $(CSRC)/pclseg_api.hip

# A/B build: the experiment switches of DESIGN.md §9/§10 (PCLSEG_GEOM, PCLSEG_DN8, ...) are read from the
# environment (PCLSEG_LIB=.../libpclseg_tuning.so); the shipped library carries only their defaults
tuning: $(CSRC)/pclseg_api.hip $(CSRC)/pclseg_kernels.h $(CSRC)/pclseg_graph.h include/pclseg.h
	$(HIPCC) $(HIPFLAGS) -DPCLSEG_TUNING -o pclsegmentation_amd/libpclseg_tuning.so $(CSRC)/pclseg_api.hip

clean:
	rm -f $(LIB) pclsegmentation_amd/libpclseg_stamps.so pclsegmentation_amd/libpclseg_tuning.so

.PHONY: all clean stamps tuning
