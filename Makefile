# Convenience targets; the Python entry points (`__graft_entry__.build()`, `python -m spectrograms_amd.build`) run the same
# commands.  The library is plain hipcc output: a Rust `build.rs` of the reference crate's `hip` feature would do the same.
HIPCC ?= hipcc
ARCH  ?= gfx950
CSRC  := spectrograms_amd/csrc
SRCS  := $(CSRC)/plan.hip $(CSRC)/fft2d.hip $(CSRC)/kernels_generic.hip $(CSRC)/kernels_r32x16.hip \
         $(CSRC)/kernels_fft2d.hip $(CSRC)/kernels_c2c1024.hip $(CSRC)/kernels_reg2d.hip $(CSRC)/kernels_q16x32.hip
LIB   := spectrograms_amd/libspectro_hip.so

.PHONY: all lib oracle test-cpu clean
all: lib oracle

lib: $(LIB)
$(LIB): $(SRCS) $(CSRC)/sgx_internal.h $(CSRC)/fft_inreg.h include/spectro_hip.h
	$(HIPCC) -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -shared -Iinclude -I$(CSRC) -o $@ $(SRCS)

oracle:
	$(MAKE) -C oracle

test-cpu: all
	python -m pytest tests -q -m "not gpu"

clean:
	rm -f $(LIB)
	$(MAKE) -C oracle clean || true
