#!/bin/bash
# One GPU-box check of everything that has to stay true: the gpu test-suite, the benches against their goldens, a fuzz.
# usage (on the GPU box, from the repo root): bash tools/verify_gpu.sh [fuzz_first_seed] [fuzz_count]
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1; tail -3 gpurun_out/pytest.log
for p in uk64m york yh_census; do
  timeout -k 10 300 python bench.py --preset $p --steps 5000 --warmup 96 --cpu-steps 0 2>/dev/null > gpurun_out/bench_$p.json
  python -c "import sys,json; d=json.load(open('gpurun_out/bench_$p.json')); print('$p', d['ms_per_step'], d.get('golden_check'), d['final_record'])"
done
timeout -k 10 300 python tools/fuzz_parity.py ${1:-2000} ${2:-200} | tail -1
