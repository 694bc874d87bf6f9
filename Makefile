# Top-level build: the HIP library behind include/aidax.h (gfx950 only), the LV2
# shell, and the CPU oracle (test infrastructure). `python -c "import
# __graft_entry__ as g; g.build()"` runs `make all`.
#
# -ffp-contract=off everywhere in the product: the fp64 biquads and fp32
# smoothers must round step by step like the reference built without FMA
# contraction; the NN kernels ask for FMAs explicitly (__builtin_fmaf).

HIPCC   ?= hipcc
ROCM    ?= /opt/rocm
CXX     := g++
ARCH    ?= gfx950
PKG     := aidadsp-lv2_amd
SRC     := $(PKG)/csrc
LIBDIR  := $(PKG)/lib
OBJDIR  := build/obj
CXXFLAGS := -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wextra -Iinclude
# -fno-slp-vectorize: clang's SLP pass pairs scalar fp32 FMAs/adds into v_pk_fma_f32 / v_pk_add_f32, which
# cost as much as the two scalar instructions on gfx950 plus the moves that build the pairs (measured:
# GRU-24 pipeline 88.5 -> 84.1 us, cfg3 478 -> 464 us, nothing slower by more than noise)
HIPFLAGS := --offload-arch=$(ARCH) $(CXXFLAGS) -fno-slp-vectorize

HOST_SRCS := $(SRC)/aidax_model.cpp $(SRC)/aidax_dsp_host.cpp $(SRC)/aidax_pack.cpp $(SRC)/aidax_pool.cpp $(SRC)/aidax_hub.cpp
HOST_OBJS := $(patsubst $(SRC)/%.cpp,$(OBJDIR)/%.o,$(HOST_SRCS))
KERN_OBJS := $(OBJDIR)/aidax_kernels.o $(OBJDIR)/aidax_stack.o $(OBJDIR)/aidax_mfma.o $(OBJDIR)/aidax_mfmalp.o $(OBJDIR)/aidax_convm.o $(OBJDIR)/aidax_quad.o
HDRS      := $(wildcard $(SRC)/*.h) include/aidax.h

LV2SO := $(PKG)/lv2/rt-neural-generic.so

all: $(LIBDIR)/libaidax_hip.so $(LV2SO) oracle

$(OBJDIR)/%.o: $(SRC)/%.cpp $(HDRS)
	@mkdir -p $(OBJDIR)
	$(CXX) $(CXXFLAGS) -I$(ROCM)/include -D__HIP_PLATFORM_AMD__ -c $< -o $@

$(OBJDIR)/%.o: $(SRC)/%.hip $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/libaidax_hip.so: $(HOST_OBJS) $(KERN_OBJS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -shared -o $@ $^

# The LV2 plugin shell (no lib prefix, like the reference's binary: rt-neural-generic/CMakeLists.txt:47)
$(LV2SO): $(PKG)/lv2/rt_neural_generic_lv2.cpp $(PKG)/lv2/lv2_min.h include/aidax.h $(LIBDIR)/libaidax_hip.so
	$(CXX) $(CXXFLAGS) -shared $< -o $@ -L$(LIBDIR) -laidax_hip -Wl,-rpath,'$$ORIGIN/../lib'

# An installable LV2 bundle (build/rt-neural-generic.lv2): generated TTL, the plugin binary linked with
# rpath $$ORIGIN, the HIP library next to it, the bundled models (tools/make_bundle.py does all of it).
bundle: $(LIBDIR)/libaidax_hip.so
	python3 tools/make_bundle.py --out build

oracle:
	$(MAKE) -s -C oracle all
	$(MAKE) -s -C oracle _ref

clean:
	rm -rf build $(LIBDIR) $(LV2SO)
	$(MAKE) -s -C oracle clean

.PHONY: all oracle bundle clean
