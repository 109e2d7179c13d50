# Builds the gfx950 library and the drop-in `simple` front end (same target name as the
# reference Makefile:6-7).  hipcc cross-compiles without a GPU.
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
PKG     := parallel-sequence-alignment_amd
LIB     := $(PKG)/lib/libpsa_hip.so
HIPFLAGS ?= -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-result -Wno-unused-value

all: $(LIB) simple oracle

$(LIB): $(PKG)/csrc/psa_hip.hip $(wildcard $(PKG)/csrc/*.h) include/psa.h
	mkdir -p $(PKG)/lib
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(PKG)/csrc/psa_hip.hip

simple: $(PKG)/host/simple_main.cpp $(LIB) include/psa.h
	$(HIPCC) -O2 -std=c++17 -o $@ $(PKG)/host/simple_main.cpp -L$(PKG)/lib -lpsa_hip -Wl,-rpath,'$$ORIGIN/$(PKG)/lib'

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf $(PKG)/lib simple
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
