# Builds the native libraries without Python (same commands as `python -m balm_amd.build`):
#   balm_amd/lib/libbalm_hip.so    HIP kernels + C ABI (include/balm_hip.h), gfx950 only
#   balm_amd/lib/libbalm_scene.so  host-only: synthetic scene generator, readers of the shipped data formats
HIPCC    ?= /opt/rocm/bin/hipcc
CXX      ?= g++
HIPFLAGS  = --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wall -Wno-unused-value -Wno-unused-result -Wno-unused-function
CSRC      = balm_amd/csrc
LIB       = balm_amd/lib
HIP_OBJS  = $(LIB)/kernels_accum.o $(LIB)/kernels_solve.o $(LIB)/kernels_build.o $(LIB)/kernels_voxel.o $(LIB)/kernels_cov.o $(LIB)/kernels_syrk_i8.o $(LIB)/balm_multi.o $(LIB)/balm_capi.o
HIP_DEPS  = $(CSRC)/balm_internal.h $(CSRC)/host_stage.h $(CSRC)/syrk_mfma_asm.inc $(CSRC)/kernels_window.inc $(CSRC)/kernels_chain.inc $(CSRC)/kernels_small.inc include/balm_hip.h

all: $(LIB)/libbalm_hip.so $(LIB)/libbalm_scene.so

$(LIB)/%.o: $(CSRC)/%.hip $(HIP_DEPS)
	@mkdir -p $(LIB)
	$(HIPCC) $(HIPFLAGS) $(if $(filter kernels_voxel,$*),-ffp-contract=off,) -c $< -o $@

$(LIB)/libbalm_hip.so: $(HIP_OBJS)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $(HIP_OBJS) -ldl -pthread

$(LIB)/libbalm_scene.so: $(CSRC)/virtual_scene.cpp $(CSRC)/readers.cpp
	@mkdir -p $(LIB)
	$(CXX) -O3 -std=c++14 -fPIC -shared -pthread -o $@ $^

$(CSRC)/syrk_mfma_asm.inc: $(CSRC)/gen/gen_syrk_asm.py
	python3 $<

clean:
	rm -f $(LIB)/*.o $(LIB)/*.so

.PHONY: all clean
