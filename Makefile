# Builds the product library (HIP, gfx950 only) and the oracle (test infrastructure).
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
PKG     := zig-lz4_amd
CSRC    := $(PKG)/csrc
LIB     := $(PKG)/libzlz4_amd.so
TUNELIB := $(PKG)/libzlz4_amd_tuning.so
HIPSRC  := $(CSRC)/zlz4_capi.hip $(CSRC)/zlz4_frame.hip $(CSRC)/zlz4_decompress.hip \
           $(CSRC)/zlz4_compress_fast.hip $(CSRC)/zlz4_compress_hc.hip $(CSRC)/zlz4_compress_hc_serial.hip
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Iinclude

all: $(LIB) $(TUNELIB) oracle

$(LIB): $(HIPSRC) $(CSRC)/zlz4_device.hpp $(CSRC)/zlz4_host.hpp include/zlz4_amd.h
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(HIPSRC)

# the same library with the experiment / A-B knobs of DESIGN.md section 7 compiled in (environment variables); used by
# tests/test_gpu_lane_decoder.py and the tools/ scripts, never by bench.py or the parity tests
tuning: $(TUNELIB)
$(TUNELIB): $(HIPSRC) $(CSRC)/zlz4_device.hpp $(CSRC)/zlz4_host.hpp include/zlz4_amd.h
	$(HIPCC) $(HIPFLAGS) -DZLZ4_TUNING -shared -o $@ $(HIPSRC)

# diagnostic build: per-phase cycle stamps inside the compress kernel (never shipped / never benchmarked)
stamps: $(HIPSRC) $(CSRC)/zlz4_device.hpp $(CSRC)/zlz4_host.hpp include/zlz4_amd.h
	$(HIPCC) $(HIPFLAGS) -DZLZ4_STAMPS -DZLZ4_TUNING -shared -o $(PKG)/libzlz4_amd_stamps.so $(HIPSRC)

oracle:
	$(MAKE) -C oracle

# audit of the decoder's hand-issued loads (see tools/check_decoder_asm.py); run after any toolchain / flag change
check-asm:
	python3 tools/check_decoder_asm.py

clean:
	rm -f $(LIB) $(TUNELIB)
	$(MAKE) -C oracle clean
.PHONY: all oracle clean stamps tuning check-asm
