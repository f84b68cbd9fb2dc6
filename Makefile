# Convenience targets; everything also runs without make (see README.md).
PY ?= python

.PHONY: build test-cpu test-gpu bench smoke clean

build:            ## hipcc --offload-arch=gfx950, in-tree libimh_hip.so (cross-compiles without a GPU)
	$(PY) -c "import __graft_entry__ as g; g.build()"

test-cpu:         ## oracle vs reference golden vectors, host logic, ABI symbols, gloo PNS (no GPU)
	$(PY) -m pytest tests -q -m "not gpu"

test-gpu:         ## MI355X parity suite through the C ABI
	$(PY) -m pytest tests -q -m gpu

smoke:
	$(PY) -c "import __graft_entry__ as g; g.smoke()"

bench:
	$(PY) bench.py --gpus 1

clean:
	rm -rf imagharmony_amd/csrc/_obj imagharmony_amd/libimh_hip.so tools/tmp_libs tests/emu/emu_layout
