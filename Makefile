# Top-level build: the HIP engine (gfx950 only) and the CPU oracle (test infrastructure).
HIPCC ?= /opt/rocm/bin/hipcc
PKG := u96-slam_amd
CSRC := $(PKG)/csrc
LIB := $(PKG)/lib/libsbm_hip.so
# --offload-compress: the device code objects are stored compressed (10.5 MB -> 2.3 MB; ~4 ms of decompression at first use)
HIPFLAGS ?= --offload-arch=gfx950 --offload-compress -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result
SRCS := $(CSRC)/sbm_api.hip $(CSRC)/sbm_prefilter.hip $(CSRC)/sbm_sad_generic.hip $(CSRC)/sbm_sad_wide.hip $(CSRC)/sbm_sad_fast.hip $(CSRC)/sbm_sad_fast_pw1.hip $(CSRC)/sbm_sad_fast_pw2.hip $(CSRC)/sbm_sad_fast_pw3.hip $(CSRC)/sbm_sad_fast_pp.hip $(CSRC)/sbm_lrcheck.hip $(CSRC)/sbm_speckle.hip $(CSRC)/sbm_consume.hip $(CSRC)/sbm_rectify.hip $(CSRC)/sbm_fpga.hip $(CSRC)/sbm_gftt.hip
OBJS := $(SRCS:.hip=.o)

all: $(LIB) oracle

FAST_HDRS := $(CSRC)/sbm_sad_fast_core.h $(CSRC)/sbm_sad_fast_strip.h $(CSRC)/sbm_sad_fast_pp_strip.h $(CSRC)/sbm_sad_fast_kernel.h $(CSRC)/sbm_sad_fast_dev.h $(CSRC)/sbm_sad_border_wave.h
$(CSRC)/sbm_sad_fast.o $(CSRC)/sbm_sad_fast_pw1.o $(CSRC)/sbm_sad_fast_pw2.o $(CSRC)/sbm_sad_fast_pw3.o $(CSRC)/sbm_sad_fast_pp.o: $(FAST_HDRS)

$(CSRC)/%.o: $(CSRC)/%.hip $(CSRC)/sbm_common.h include/sbm.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	@mkdir -p $(PKG)/lib
	$(HIPCC) --offload-arch=gfx950 --offload-compress -shared -fPIC -o $@ $(OBJS)

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(OBJS) $(LIB)
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
