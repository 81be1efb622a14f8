#!/bin/bash
# One GPU-box check of everything that has to stay true: the gpu test-suite, a fuzz, the benches against their goldens.
# usage (on the GPU box, from the repo root): bash tools/verify_gpu.sh tag [fuzz_first_seed] [fuzz_count] [pytest -k expression]
tag=${1:-x}
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q -k "${4:-not nothing}" > gpurun_out/pytest_$tag.log 2>&1; tail -4 gpurun_out/pytest_$tag.log
timeout -k 10 300 python tools/fuzz_parity.py ${2:-2000} ${3:-200} > gpurun_out/fuzz_$tag.log 2>&1; tail -1 gpurun_out/fuzz_$tag.log
timeout -k 10 200 python tools/fuzz_parity.py 9000 12 --big > gpurun_out/fuzz_big_$tag.log 2>&1; tail -1 gpurun_out/fuzz_big_$tag.log
ESIM_GRID_CHUNK=16 timeout -k 10 300 python tools/fuzz_parity.py 12000 30 --big > gpurun_out/fuzz_g16_$tag.log 2>&1; tail -1 gpurun_out/fuzz_g16_$tag.log
ESIM_GRID_CHUNK=48 timeout -k 10 300 python tools/fuzz_parity.py 13000 30 --big > gpurun_out/fuzz_g48_$tag.log 2>&1; tail -1 gpurun_out/fuzz_g48_$tag.log
