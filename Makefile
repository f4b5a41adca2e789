# Convenience targets; everything here is a one-line wrapper of a command documented in README.md / DESIGN.md.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950

.PHONY: build test-cpu test-gpu bench selftest microbench resources clean

build:                       ## libmoshi_mi.so (hipcc, gfx950; cross-compiles without a GPU)
	python -m moshi_amd.build

test-cpu:                    ## oracle vs golden vectors, host logic, the kernels on the CPU simulator, C-ABI checks (~5 min)
	python -m pytest tests -x -q -m "not gpu"

test-gpu:                    ## parity on a real MI355X through the C ABI (~10 min)
	python -m pytest tests -x -q -m gpu

bench:                       ## the headline line (1 GPU); `python bench.py --gpus N` for N ranks
	python bench.py

build/native_selftest: scripts/native_selftest.cpp include/moshi_mi.h build
	mkdir -p build
	$(HIPCC) --offload-arch=$(ARCH) -O2 -std=c++17 -ffp-contract=off -Iinclude scripts/native_selftest.cpp -Lmoshi_amd -lmoshi_mi \
	    -Wl,-rpath,'$$ORIGIN/../moshi_amd' -o $@

selftest: build/native_selftest   ## Python-free check of the library against the oracle's recorded outputs (~1 s on the GPU box)
	build/native_selftest tests/golden/native_selftest
	build/native_selftest tests/golden/native_selftest_b18

microbench:                  ## the product's GEMM kernels on the 7B shapes + the L2 access-pattern probe (native, seconds)
	mkdir -p build
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -Imoshi_amd/csrc scripts/gemm_microbench.hip moshi_amd/csrc/api_common.hip -o build/gemm_microbench
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 scripts/l2_pattern_probe.hip -o build/l2_pattern_probe

resources:                   ## per-kernel register / LDS / scratch table from the built library (no GPU)
	python scripts/kernel_resources.py

clean:
	rm -rf build moshi_amd/csrc/*.o moshi_amd/libmoshi_mi.so tests/hipsim/*.o tests/hipsim/libmoshi_sim.so
