# Builds the gfx950 library and the drop-in `simple` front end (same target name as the
# reference Makefile:6-7).  hipcc cross-compiles without a GPU.
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
PKG     := parallel-sequence-alignment_amd
LIB     := $(PKG)/lib/libpsa_hip.so
HIPFLAGS ?= -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-result -Wno-unused-value

all: $(LIB) simple oracle

# One object per translation unit: a change to one kernel family recompiles only that family (make -j builds them
# side by side; the fill kernels with and without winner codes are the two long ones).
CS      := $(PKG)/csrc
OBJDIR  := build/obj
OBJS    := $(addprefix $(OBJDIR)/,psa_hip.o psa_fill_codes.o psa_fill_score.o psa_fill_pk.o psa_trace.o psa_batch.o)
COMMON  := $(CS)/psa_args.hip.h $(CS)/psa_device.hip.h $(CS)/psa_launch.h $(CS)/psa_host.h include/psa.h
FILLH   := $(CS)/psa_fill_tu.hip.h $(CS)/psa_fill_v2.hip.h $(CS)/psa_fill_v1.hip.h

$(OBJDIR)/psa_hip.o: $(CS)/psa_hip.hip $(COMMON)
$(OBJDIR)/psa_fill_codes.o: $(CS)/psa_fill_codes.hip $(FILLH) $(COMMON)
$(OBJDIR)/psa_fill_score.o: $(CS)/psa_fill_score.hip $(FILLH) $(CS)/psa_fill_aux.hip.h $(COMMON)
$(OBJDIR)/psa_fill_pk.o: $(CS)/psa_fill_pk.hip $(CS)/psa_fill_pk.hip.h $(FILLH) $(COMMON)
$(OBJDIR)/psa_trace.o: $(CS)/psa_trace.hip $(CS)/psa_trace1.hip.h $(CS)/psa_trace2.hip.h $(COMMON)
$(OBJDIR)/psa_batch.o: $(CS)/psa_batch.hip $(CS)/psa_batch.hip.h $(COMMON)

PKWAVES ?= 3
$(OBJDIR)/psa_fill_pk.o: HIPFLAGS += -DPSA_PK_WAVES=$(PKWAVES) $(PKEXTRA)
$(OBJDIR)/psa_trace.o: HIPFLAGS += $(TBEXTRA)
# (strips per workgroup of the packed kernel: host and kernel must agree, so for every object)
ifdef PKNW
HIPFLAGS += -DPSA_PK_NW=$(PKNW)
endif

$(OBJDIR)/%.o:
	mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

$(LIB): $(OBJS)
	mkdir -p $(PKG)/lib
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(OBJS)

simple: $(PKG)/host/simple_main.cpp $(LIB) include/psa.h
	$(HIPCC) -O2 -std=c++17 -o $@ $(PKG)/host/simple_main.cpp -L$(PKG)/lib -lpsa_hip -Wl,-rpath,'$$ORIGIN/$(PKG)/lib'

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf $(PKG)/lib simple $(OBJDIR)
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
