# Top-level convenience Makefile (C users; Python users call __graft_entry__.build()).
#   make            libmlkem_amd.so (HIP kernels + batch C-ABI) and libml_kem.so (ml_kem.h drop-in shim)
#   make checkers   oracle/liboracle_mlkem.so (+ oracle/_ref/ when the reference tree is present) and the host wave emulator
#   make test       CPU test tier            make gpu-test   GPU test tier (needs an MI355X)
HIPCC ?= $(shell command -v hipcc 2>/dev/null || echo /opt/rocm/bin/hipcc)
PKG   := crystals-kyber_amd
CSRC  := $(PKG)/csrc
HDRS  := $(wildcard $(CSRC)/*.hpp) include/mlkem_batch.h include/mlkem_compat.h

all: $(PKG)/libmlkem_amd.so $(PKG)/libml_kem.so

$(PKG)/libmlkem_amd.so: $(CSRC)/mlkem_capi.hip $(CSRC)/exports.map $(HDRS)
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fvisibility=hidden -fPIC -shared -Wl,--version-script=$(CSRC)/exports.map -o $@ $(CSRC)/mlkem_capi.hip

$(PKG)/libml_kem.so: $(CSRC)/ml_kem_shim.c $(PKG)/libmlkem_amd.so $(HDRS)
	gcc -O2 -fPIC -shared -o $@ $(CSRC)/ml_kem_shim.c -L$(PKG) -lmlkem_amd -Wl,-rpath,'$$ORIGIN'

checkers:
	$(MAKE) -s -C oracle all
	g++ -O2 -std=c++17 -pthread -fPIC -shared -Wno-unknown-pragmas -Wno-attributes -o tests/emu/libmlkem_emu.so tests/emu/emu_lib.cpp

test: all checkers
	python -m pytest tests -q -m "not gpu"

gpu-test: all checkers
	python -m pytest tests -q -m gpu

clean:
	rm -f $(PKG)/libmlkem_amd.so $(PKG)/libml_kem.so tests/emu/libmlkem_emu.so
	$(MAKE) -s -C oracle clean

.PHONY: all checkers test gpu-test clean
