# make            -> libwinograd_mi355x.so + ./Test (the reference's `make && ./Test 0` UX)
# make oracle     -> oracle/liboracle.so (CPU checker, test infrastructure only)
# hipcc cross-compiles for gfx950 without a GPU.
HIPCC   ?= hipcc
CC      ?= gcc
ARCH    ?= gfx950
PKG     := cuda-winograd_amd
CSRC    := $(PKG)/csrc
HOST    := $(PKG)/host
BUILD   := build

HIPFLAGS := --offload-arch=$(ARCH) -O3 -fPIC -std=c++17 -Iinclude -I$(CSRC) -Wall -Wno-unused-function
CFLAGS   := -O2 -fPIC -std=gnu11 -Iinclude -I$(HOST) -Wall

HIP_SRCS := $(wildcard $(CSRC)/*.hip)
HIP_OBJS := $(patsubst $(CSRC)/%.hip,$(BUILD)/%.o,$(HIP_SRCS))
HOST_OBJS := $(BUILD)/util.o $(BUILD)/layer_driver.o $(BUILD)/cpu_baseline.o

LIB := $(PKG)/libwinograd_mi355x.so

all: $(LIB) Test

$(BUILD):
	mkdir -p $(BUILD)

$(BUILD)/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.h) include/winograd_mi355x.h | $(BUILD)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(BUILD)/cpu_baseline.o: CFLAGS := -O3 -march=x86-64-v3 -fPIC -std=gnu11 -Iinclude -I$(HOST) -Wall
$(BUILD)/%.o: $(HOST)/%.c $(wildcard include/*.h) $(wildcard $(HOST)/*.h) | $(BUILD)
	$(CC) $(CFLAGS) -c $< -o $@

$(LIB): $(HIP_OBJS) $(HOST_OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^ -lpthread -lm

Test: $(BUILD)/Test.o $(LIB)
	$(CC) -o $@ $(BUILD)/Test.o -L$(PKG) -lwinograd_mi355x -Wl,-rpath,'$$ORIGIN/$(PKG)' -lpthread -lm

oracle: oracle/liboracle.so
oracle/liboracle.so: oracle/cpu_conv.c
	$(CC) -O3 -march=x86-64-v3 -fPIC -shared -std=gnu11 -o $@ $< -lpthread -lm

# developer tools (ablation of the fused kernel, fp32 MFMA ceiling); not part of the library
tools: tools/ablate_fused tools/ablate_1x1 tools/mfma_peak tools/coissue tools/mixbench tools/occ1x1
tools/ablate_1x1: tools/ablate_1x1.hip $(wildcard $(CSRC)/*.h)
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -Iinclude -I$(CSRC) $< -o $@
tools/ablate_fused: tools/ablate_fused.hip $(wildcard $(CSRC)/*.h)
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -Iinclude -I$(CSRC) $< -o $@
# experiment variants of the ablation tool: make tools/xab_NAME XFLAGS="-DWINO_DMA_MODE=1"
tools/xab_%: tools/ablate_fused.hip $(wildcard $(CSRC)/*.h)
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -Iinclude -I$(CSRC) $(XFLAGS) tools/ablate_fused.hip -o $@
tools/coissue: tools/coissue.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 $< -o $@
tools/mfma_peak: tools/mfma_peak.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 $< -o $@
tools/mixbench: tools/mixbench.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 $< -o $@
tools/occ1x1: tools/occ1x1.hip $(wildcard $(CSRC)/*.h)
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -Iinclude -I$(CSRC) $< -o $@

clean:
	rm -rf $(BUILD) $(LIB) Test oracle/liboracle.so tools/ablate_fused tools/ablate_1x1 tools/mfma_peak tools/coissue tools/mixbench tools/occ1x1

.PHONY: all oracle tools clean
