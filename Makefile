# Builds the MI355X match-search library, the host library and the drop-in
# program (same target name as the reference's Makefile: gpm-par.exe).
#
#   make            libgpm_hip.so, libgpm_host.so, gpm-par.exe  (in mpi_amd/lib, repo root)
#   make sanitize   host C code and the oracle under ASan + UBSan (mpi_amd/lib/san/)
#   make oracle     CPU oracle (test infrastructure) and, if possible, the reference build
HIPCC   ?= /opt/rocm/bin/hipcc
CC      ?= gcc
ARCH    ?= gfx950
CFLAGS  ?= -O2 -std=c99 -Wall -Wextra -fPIC
HIPFLAGS ?= -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -Wall -Wno-unused-result
LIBDIR  := mpi_amd/lib
HOST_SRC := mpi_amd/host/utils.c mpi_amd/host/graph.c mpi_amd/host/loader.c mpi_amd/host/plan.c
HOST_HDR := mpi_amd/host/utils.h mpi_amd/host/graph.h mpi_amd/host/loader.h mpi_amd/host/plan.h include/gpm_hip.h

all: $(LIBDIR)/libgpm_hip.so $(LIBDIR)/libgpm_host.so gpm-par.exe

$(LIBDIR)/libgpm_hip.so: mpi_amd/csrc/gpm_hip.hip mpi_amd/csrc/gpm_kernels.hpp mpi_amd/csrc/gpm_device.hpp include/gpm_hip.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ mpi_amd/csrc/gpm_hip.hip -ldl -lpthread

$(LIBDIR)/libgpm_host.so: $(HOST_SRC) $(HOST_HDR)
	@mkdir -p $(LIBDIR)
	$(CC) $(CFLAGS) -shared -o $@ $(HOST_SRC) -lpthread

gpm-par.exe: mpi_amd/host/gpm-par.c $(HOST_SRC) $(HOST_HDR) $(LIBDIR)/libgpm_hip.so
	$(CC) $(CFLAGS) -o $@ mpi_amd/host/gpm-par.c $(HOST_SRC) -L$(LIBDIR) -lgpm_hip -lpthread -Wl,-rpath,'$$ORIGIN/$(LIBDIR)'

# The host-side C (parser, threaded loader, CSR cache, planner) and the oracle under AddressSanitizer
# and UndefinedBehaviorSanitizer, next to a plain build of the same checker; CPU code only.
SANFLAGS ?= -O1 -g -std=c99 -Wall -Wextra -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer
sanitize: $(LIBDIR)/san/host_check $(LIBDIR)/san/host_check_plain $(LIBDIR)/san/gpm_oracle

$(LIBDIR)/san/host_check: mpi_amd/host/host_check.c $(HOST_SRC) $(HOST_HDR)
	@mkdir -p $(LIBDIR)/san
	$(CC) $(SANFLAGS) -o $@ mpi_amd/host/host_check.c $(HOST_SRC) -lpthread

$(LIBDIR)/san/host_check_plain: mpi_amd/host/host_check.c $(HOST_SRC) $(HOST_HDR)
	@mkdir -p $(LIBDIR)/san
	$(CC) -O2 -std=c99 -Wall -Wextra -o $@ mpi_amd/host/host_check.c $(HOST_SRC) -lpthread

$(LIBDIR)/san/gpm_oracle: oracle/gpm_oracle.c oracle/gpm_oracle.h
	@mkdir -p $(LIBDIR)/san
	$(CC) $(SANFLAGS) -DORACLE_MAIN -o $@ oracle/gpm_oracle.c

oracle:
	$(MAKE) -C oracle all
	-$(MAKE) -C oracle ref

clean:
	rm -rf $(LIBDIR) gpm-par.exe

.PHONY: all oracle clean sanitize
