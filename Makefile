# Top-level build: the HIP engine (libspmv_hip.so), the C++ compat shim + harness, and the oracle.
#   make            -> engine + compat + harness + oracle restatement
#   make engine     -> arm-spmv_amd/lib/libspmv_hip.so       (hipcc, gfx950 only)
#   make oracle     -> oracle/_build/libspmv_oracle.so     (test infrastructure)
#   make ref        -> oracle/_ref/libarmspmv_ref.so       (needs /root/reference; build container only)
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
PKG      := arm-spmv_amd
CSRC     := $(PKG)/csrc
LIBDIR   := $(PKG)/lib
OBJDIR   := build/obj
HIPFLAGS := -O3 -Wno-unused-result -Wno-unused-value -std=c++17 -fPIC --offload-arch=$(ARCH) -Iinclude -I$(CSRC) -Wall -Wno-unused-function -ffp-contract=on

ENGINE_SRCS := $(wildcard $(CSRC)/*.hip)
ENGINE_OBJS := $(patsubst $(CSRC)/%.hip,$(OBJDIR)/%.o,$(ENGINE_SRCS))

all: engine host oracle

engine: $(LIBDIR)/libspmv_hip.so

$(OBJDIR)/%.o: $(CSRC)/%.hip $(CSRC)/common.hpp $(CSRC)/wave.hpp $(CSRC)/placement_math.hpp $(CSRC)/panel_groups.hpp $(CSRC)/split_rows.hpp include/spmv_abi.h
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

# The link is gated: no kernel on a measured path may use scratch memory or spill registers (tools/hot_kernels.txt lists
# them; the check reads the code objects' metadata notes, no GPU needed).  `make engine NOSCRATCH_GATE=0` links anyway.
NOSCRATCH_GATE ?= 1
$(LIBDIR)/libspmv_hip.so: $(ENGINE_OBJS) tools/hot_kernels.txt tools/kernel_resources.py
	@mkdir -p $(LIBDIR)
	@if [ "$(NOSCRATCH_GATE)" = "1" ]; then python3 tools/kernel_resources.py --check tools/hot_kernels.txt $(ENGINE_OBJS) > $(OBJDIR)/kernel_resources.txt \
	    || { cat $(OBJDIR)/kernel_resources.txt; echo "make engine: a hot kernel uses scratch (see above)"; exit 1; }; tail -1 $(OBJDIR)/kernel_resources.txt; fi
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $(ENGINE_OBJS)

host: engine
	@if [ -f $(PKG)/host/Makefile ]; then $(MAKE) -C $(PKG)/host; fi

oracle:
	$(MAKE) -C oracle

ref:
	$(MAKE) -C oracle ref

clean:
	rm -rf build $(LIBDIR)
	$(MAKE) -C oracle clean

.PHONY: all engine host oracle ref clean
