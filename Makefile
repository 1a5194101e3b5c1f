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

# host code under AddressSanitizer + UBSan (sanitizers on the host side only; one fused shape keeps the build short), linked
# with a driver that walks the C ABI without a GPU: build/abi_asan  (tests/test_host_asan_cpu.py runs it)
ASANFLAGS := -O1 -g -std=c++17 -fPIC --offload-arch=$(ARCH) -Wno-unused-function -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined \
	-Xarch_host -fno-sanitize-recover=undefined '-DFT_SHAPES(X)=X(10,10,40,50)'
ASANOBJS := $(patsubst $(CSRC)/%.hip,build/asan_%.o,$(SRCS))
build/asan_%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p build
	$(HIPCC) $(ASANFLAGS) -c $< -o $@
build/abi_asan: $(ASANOBJS) tests/host/abi_asan.c
	gcc -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -c tests/host/abi_asan.c -o build/abi_asan_main.o
	$(HIPCC) --offload-arch=$(ARCH) -fsanitize=address,undefined -fno-gpu-sanitize build/abi_asan_main.o $(ASANOBJS) -o $@
asan: build/abi_asan

clean:
	rm -f $(OBJS) $(LIB)

.PHONY: all clean diag asan
