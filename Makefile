# Build recipes.  `make` builds everything that can be built on a machine
# without a GPU: the host library, the HIP library (cross-compiled for gfx950),
# the command-line front end and the test oracle.
ROCM     ?= /opt/rocm
HIPCC    ?= $(ROCM)/bin/hipcc
CXX      ?= g++
CC       ?= gcc
PKG      := pathtracing_amd
HOSTSRC  := $(wildcard $(PKG)/csrc/host/*.cpp)
DEVDIR   := $(PKG)/csrc/device
DEVOBJ   := $(DEVDIR)/pt_abi.o $(DEVDIR)/pt_exact.o $(DEVDIR)/pt_fast.o $(DEVDIR)/pt_refine.o
DEVHDR   := $(wildcard $(PKG)/csrc/device/*.h) $(wildcard include/*.h)
CXXFLAGS := -O2 -std=c++17 -fPIC -Wall -Wextra -ffp-contract=off -Iinclude
HIPFLAGS := -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Wall
# The fp64 reference-semantics kernels must round after every operation; the
# fp32 GI kernels may fuse multiply-adds.
$(DEVDIR)/pt_abi.o $(DEVDIR)/pt_exact.o $(DEVDIR)/pt_refine.o: CONTRACT := -ffp-contract=off
$(DEVDIR)/pt_fast.o: CONTRACT := -ffp-contract=fast

all: host device cli oracle tools

host: $(PKG)/libpthost.so
device: $(PKG)/libptamd.so
cli: $(PKG)/ptmain
oracle: oracle/libptoracle.so
tools: tools/ta_rate

$(PKG)/libpthost.so: $(HOSTSRC) $(PKG)/csrc/host/scene.hpp include/pathtracing_host.h include/pathtracing_amd.h
	$(CXX) $(CXXFLAGS) -shared -o $@ $(HOSTSRC)

$(DEVDIR)/%.o: $(DEVDIR)/%.hip $(DEVHDR)
	$(HIPCC) $(HIPFLAGS) $(CONTRACT) -c -o $@ $<

$(PKG)/libptamd.so: $(DEVOBJ)
	$(HIPCC) --offload-arch=gfx950 -shared -o $@ $(DEVOBJ)

$(PKG)/ptmain: $(PKG)/csrc/cli/ptmain.cpp $(PKG)/libpthost.so $(PKG)/libptamd.so
	$(CXX) $(CXXFLAGS) -o $@ $< -pthread -L$(PKG) -lpthost -lptamd -Wl,-rpath,'$$ORIGIN' -L$(ROCM)/lib -Wl,-rpath,$(ROCM)/lib

# load-rate micro-benchmark that bench.py runs beside its timings (the physical ceiling of the traversal kernel)
tools/ta_rate: tools/ta_rate.hip
	$(HIPCC) -O3 --offload-arch=gfx950 -o $@ $<

oracle/libptoracle.so: oracle/pt_oracle.c
	$(MAKE) -C oracle

# Sanitizer builds of the CPU code (host library, oracle; the test helpers follow PT_SANITIZE=1 themselves) and the CPU
# tests that feed them untrusted bytes (texture decoders, packed scenes), the kd build and the search rules, run under them.
# The Python interpreter is not instrumented, so the AddressSanitizer runtime is put in front of whatever is preloaded already.
SAN      := -fsanitize=address,undefined -fno-sanitize-recover=all -fno-omit-frame-pointer -g
ASAN_DIR := build/asan
asan: $(ASAN_DIR)/libpthost.so $(ASAN_DIR)/libptoracle.so
$(ASAN_DIR)/libpthost.so: $(HOSTSRC) $(PKG)/csrc/host/scene.hpp include/pathtracing_host.h include/pathtracing_amd.h
	mkdir -p $(ASAN_DIR)
	$(CXX) $(CXXFLAGS) -O1 $(SAN) -shared -o $@ $(HOSTSRC)
$(ASAN_DIR)/libptoracle.so: oracle/pt_oracle.c
	mkdir -p $(ASAN_DIR)
	$(CC) -O1 -std=gnu11 -fPIC -shared -fopenmp -ffp-contract=off $(SAN) -o $@ oracle/pt_oracle.c -lm
ASAN_TESTS ?= tests/test_host.py tests/test_refine_host.py tests/test_research_rule.py tests/test_packet_rule.py tests/test_oracle_pins.py tests/test_golden.py
asan-test: asan
	PT_SANITIZE=1 ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
	LD_PRELOAD="$$($(CC) -print-file-name=libasan.so):$$LD_PRELOAD" python3 -m pytest $(ASAN_TESTS) -q -m "not gpu" -p no:cacheprovider

clean:
	rm -rf build
	rm -f $(DEVOBJ) $(PKG)/libpthost.so $(PKG)/libptamd.so $(PKG)/ptmain oracle/libptoracle.so tools/ta_rate

.PHONY: all host device cli oracle tools clean asan asan-test
