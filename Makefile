# Builds libivae_amd.so (HIP kernels + C ABI) for gfx950.  hipcc cross-compiles without a GPU.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := ivae_amd/csrc
SRCS := $(wildcard $(CSRC)/*.hip)
OBJS := $(SRCS:.hip=.o)
HDRS := $(wildcard $(CSRC)/*.h) include/ivae_amd.h
LIB := ivae_amd/libivae_amd.so
CXXFLAGS := -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-function

all: $(LIB)

$(CSRC)/%.o: $(CSRC)/%.hip $(HDRS)
	$(HIPCC) $(CXXFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) $(OBJS) -o $@

# diagnostic build with in-kernel phase stamps: another library name, -DIVAE_DIAG (the sources refuse their diagnostic
# switches without it; ivae_version() says so) -- never shipped, never benchmarked
diag: ivae_amd/libivae_amd_diag.so
ivae_amd/libivae_amd_diag.so: $(SRCS) $(HDRS)
	$(HIPCC) $(CXXFLAGS) -DIVAE_DIAG -DFT_PROF $(DIAGFLAGS) -shared $(SRCS) -o $@

clean:
	rm -f $(OBJS) $(LIB)

.PHONY: all clean diag
