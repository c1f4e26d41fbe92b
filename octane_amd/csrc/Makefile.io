# Builds the file layer of the reference's CLI on this library (SURVEY 8f N3):
#   liboctane_io.so  -- oct_fileread (GOES / polar / Mercator / CLAVR-x / first guess) and oct_filewrite (outfile*.nc) on nc4lite (HDF5)
#   octane           -- the command line: octane -i1 a.nc -i2 b.nc [-o outdir/] ...
# Needs an HDF5 >= 1.10 with the high-level library (H5DS).  This image has one under /opt/conda only; where none is
# found the targets are skipped (the flow library itself does not depend on any of this).
# Link notes: the HDF5 libraries are named by path and only liboctane_io.so carries a RUNPATH to them -- putting
# $(HDF5_ROOT)/lib on the link or run path of the programs would also pick up that tree's (older) libstdc++.
CXX       ?= g++
HDF5_ROOT ?= /opt/conda
HERE  := $(dir $(abspath $(lastword $(MAKEFILE_LIST))))
# SAN=1: AddressSanitizer + UndefinedBehaviorSanitizer build into octane_amd/_san/ (after `make -f Makefile.host SAN=1`)
SAN   ?= 0
ifeq ($(SAN),1)
LIBD  := $(HERE)../_san
OPT   := -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined
else
LIBD  := $(HERE)..
OPT   := -O2
endif
HAVE_HDF5 := $(wildcard $(HDF5_ROOT)/include/hdf5_hl.h)

ifeq ($(HAVE_HDF5),)
all:
	@echo "no HDF5 under $(HDF5_ROOT): skipping liboctane_io.so and the octane command line"
else
all: $(LIBD)/liboctane_io.so $(LIBD)/octane

$(LIBD)/liboctane_io.so: $(HERE)io/nc4lite.cpp $(HERE)io/goes_io.cpp $(HERE)io/nc4lite.hpp $(LIBD)/liboctane_host.so
	$(CXX) $(OPT) -fPIC -shared -std=c++17 -Wall -I$(HDF5_ROOT)/include -o $@ $(HERE)io/nc4lite.cpp $(HERE)io/goes_io.cpp \
	    -L$(LIBD) -loctane_host -loctane_vof $(HDF5_ROOT)/lib/libhdf5_hl.so $(HDF5_ROOT)/lib/libhdf5.so \
	    -Wl,--enable-new-dtags -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,$(HDF5_ROOT)/lib

$(LIBD)/octane: $(HERE)io/octane_main.cpp $(LIBD)/liboctane_io.so
	$(CXX) $(OPT) -std=c++17 -Wall -o $@ $(HERE)io/octane_main.cpp -L$(LIBD) -loctane_io -loctane_host -loctane_vof \
	    -Wl,-rpath,'$$ORIGIN' -Wl,-rpath-link,$(HDF5_ROOT)/lib
endif

clean:
	rm -f $(LIBD)/liboctane_io.so $(LIBD)/octane
.PHONY: all clean
