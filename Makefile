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

$(OBJDIR)/%.o: $(CSRC)/%.hip $(CSRC)/common.hpp $(CSRC)/wave.hpp include/spmv_abi.h
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/libspmv_hip.so: $(ENGINE_OBJS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $^

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
