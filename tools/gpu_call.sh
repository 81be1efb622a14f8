#!/bin/bash
# One GPU-box call of this round: steps joined so that a failing GPU step ends the call (no GPU step after a fault).
# usage: bash tools/gpu_call.sh tag step...   steps: tests calib prof bench
tag=$1; shift
mkdir -p gpurun_out
steps="$@"
for step in $steps; do
  case $step in
    tests) timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_$tag.log 2>&1; rc=$?; tail -5 gpurun_out/pytest_$tag.log; [ $rc -eq 0 ] || exit $rc ;;
    calib) timeout -k 10 300 bash tools/calib_passes.sh > gpurun_out/calib_$tag.log 2>&1; rc=$?; tail -12 gpurun_out/calib_$tag.log; cp profiles/calib_tcc_calibration.* gpurun_out/ 2>/dev/null; [ $rc -eq 0 ] || exit $rc ;;
    prof) timeout -k 10 900 bash tools/profile_passes_r03.sh $tag uk64m > gpurun_out/prof_$tag.log 2>&1; rc=$?; tail -70 gpurun_out/prof_$tag.log; cp profiles/${tag}_* gpurun_out/ 2>/dev/null; [ $rc -eq 0 ] || exit $rc ;;
    bench) [ -f profiles/${tag}_summary.json ] && cp profiles/${tag}_summary.json profiles/current_uk64m.json
           timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err; rc=$?; cut -c1-2500 gpurun_out/bench_$tag.json; [ $rc -eq 0 ] || { tail -5 gpurun_out/bench_$tag.err; exit $rc; } ;;
    ab) for i in 1 2; do
          for m in "1 1" "2 2" "4 4" "4 1" "1 4" "8 8"; do set -- $m; echo -n "draw x$1 units x$2: "; ESIM_DRAW_MULT=$1 ESIM_UNITS_MULT=$2 timeout -k 10 200 python tools/run_preset.py uk64m | grep us/step | cut -c1-120 || exit 5; done
        done ;;
    wave) timeout -k 10 300 python tools/wave_profile.py uk64m 2880 3360 3840 > gpurun_out/wave_$tag.log 2>&1; rc=$?; tail -40 gpurun_out/wave_$tag.log; [ $rc -eq 0 ] || exit $rc
          timeout -k 10 300 python tools/wave_profile_units.py uk64m 2880 3360 3840 > gpurun_out/waveu_$tag.log 2>&1; rc=$?; tail -12 gpurun_out/waveu_$tag.log; [ $rc -eq 0 ] || exit $rc ;;
    fuzz) timeout -k 10 300 python tools/fuzz_parity.py 2000 80 > gpurun_out/fuzz_$tag.log 2>&1; rc=$?; tail -2 gpurun_out/fuzz_$tag.log; [ $rc -eq 0 ] || exit $rc
          ESIM_GRID_CHUNK=16 timeout -k 10 300 python tools/fuzz_parity.py 12000 20 --big > gpurun_out/fuzz_g16_$tag.log 2>&1; rc=$?; tail -1 gpurun_out/fuzz_g16_$tag.log; [ $rc -eq 0 ] || exit $rc ;;
    grid16) echo -n "grid 1024: "; timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-extra-runs --cpu-seconds 0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); t=d['config']['timed_region']; print(t['wall_us_per_step']*20, t['chunk_passes_device_ms']*1e3)" || exit 6
            echo -n "grid 16:   "; ESIM_GRID_CHUNK=16 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-extra-runs --cpu-seconds 0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); t=d['config']['timed_region']; print(t['wall_us_per_step']*20, t['chunk_passes_device_ms']*1e3)" || exit 6
            echo -n "grid 64:   "; ESIM_GRID_CHUNK=64 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-extra-runs --cpu-seconds 0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); t=d['config']['timed_region']; print(t['wall_us_per_step']*20, t['chunk_passes_device_ms']*1e3)" || exit 6
            echo -n "york grid 1024: "; timeout -k 10 300 python tools/run_preset.py york | grep us/step | cut -c1-140
            echo -n "york grid 64:   "; ESIM_GRID_CHUNK=64 timeout -k 10 300 python tools/run_preset.py york | grep us/step | cut -c1-140
            echo -n "york grid 16:   "; ESIM_GRID_CHUNK=16 timeout -k 10 300 python tools/run_preset.py york | grep us/step | cut -c1-140 ;;
    pmap) ESIM_PMAP_REBUILD=1 timeout -k 10 300 python tools/fuzz_parity.py 2000 40 > gpurun_out/fuzz_pm1_$tag.log 2>&1; rc=$?; tail -3 gpurun_out/fuzz_pm1_$tag.log; [ $rc -eq 0 ] || exit $rc
          timeout -k 10 300 python tools/fuzz_parity.py 2000 40 > gpurun_out/fuzz_pm4_$tag.log 2>&1; rc=$?; tail -3 gpurun_out/fuzz_pm4_$tag.log; [ $rc -eq 0 ] || exit $rc
          ESIM_PMAP_REBUILD=1000 timeout -k 10 300 python tools/fuzz_parity.py 3000 40 > gpurun_out/fuzz_pmx_$tag.log 2>&1; rc=$?; tail -3 gpurun_out/fuzz_pmx_$tag.log; [ $rc -eq 0 ] || exit $rc ;;
    pmab) for i in 1 2; do
            echo -n "rebuilt per chunk (old): "; ESIM_PMAP=0 timeout -k 10 200 python tools/run_preset.py uk64m | grep us/step | cut -c1-110 || exit 5
            for r in 1 2 4 8 1000; do echo -n "persistent, rebuild every $r: "; ESIM_PMAP_REBUILD=$r timeout -k 10 200 python tools/run_preset.py uk64m | grep us/step | cut -c1-110 || exit 5; done
          done
          for p in york syn3m5 yh_census; do echo -n "$p old: "; ESIM_PMAP=0 timeout -k 10 200 python tools/run_preset.py $p | grep us/step | cut -c1-110; echo -n "$p persistent: "; timeout -k 10 200 python tools/run_preset.py $p | grep us/step | cut -c1-110; done ;;
    bigfuzz) timeout -k 10 400 python tools/fuzz_parity.py 9000 16 --big > gpurun_out/fuzz_big_$tag.log 2>&1; rc=$?; tail -2 gpurun_out/fuzz_big_$tag.log; [ $rc -eq 0 ] || exit $rc
          ESIM_PMAP_REBUILD=1000 timeout -k 10 400 python tools/fuzz_parity.py 9100 16 --big > gpurun_out/fuzz_bigx_$tag.log 2>&1; rc=$?; tail -2 gpurun_out/fuzz_bigx_$tag.log; [ $rc -eq 0 ] || exit $rc
          ESIM_GRID_CHUNK=16 ESIM_PMAP_REBUILD=3 timeout -k 10 300 python tools/fuzz_parity.py 12000 20 --big > gpurun_out/fuzz_g16_$tag.log 2>&1; rc=$?; tail -1 gpurun_out/fuzz_g16_$tag.log; [ $rc -eq 0 ] || exit $rc ;;
    kt) timeout -k 10 200 python tools/kernel_times.py uk64m || exit 5 ;;
    ktgrid) for v in "1024 4" "768 4" "512 8"; do set -- $v; echo -n "grid $1 draw mult $2: "; ESIM_GRID_CHUNK=$1 ESIM_DRAW_MULT=$2 ESIM_UNITS_MULT=$2 timeout -k 10 200 python tools/kernel_times.py uk64m | cut -c40-330 || exit 5; done ;;
    ktimes) ESIM_PMAP=0 timeout -k 10 200 python tools/kernel_times.py uk64m || exit 5
            ESIM_PMAP_REBUILD=1 timeout -k 10 200 python tools/kernel_times.py uk64m || exit 5
            ESIM_PMAP_REBUILD=4 timeout -k 10 200 python tools/kernel_times.py uk64m || exit 5 ;;
    wavepm) ESIM_PMAP=0 ESIM_DRAW_MULT=1 ESIM_UNITS_MULT=1 timeout -k 10 300 python tools/wave_profile.py uk64m 2880 3840 2>&1 | grep -E "draw:|marks" | cut -c1-330
            ESIM_PMAP_REBUILD=4 ESIM_DRAW_MULT=1 ESIM_UNITS_MULT=1 timeout -k 10 300 python tools/wave_profile.py uk64m 2880 3840 2>&1 | grep -E "draw:|marks" | cut -c1-330 ;;
    counts) ESIM_PMAP=0 timeout -k 10 300 python tools/work_counts.py cnt_old uk64m | cut -c1-900 || exit 5
            timeout -k 10 300 python tools/work_counts.py cnt_pm uk64m | cut -c1-900 || exit 5 ;;
    waveu) ESIM_PMAP=0 timeout -k 10 300 python tools/wave_profile_units.py uk64m 2880 3840 2>&1 | cut -c1-330
           timeout -k 10 300 python tools/wave_profile_units.py uk64m 2880 3840 2>&1 | cut -c1-330 ;;
    presets) for p in york syn3m5 yh_census uk64m; do timeout -k 10 200 python tools/run_preset.py $p | grep us/step | cut -c1-150 || exit 5; done ;;
    smallgrid) for v in "0 1" "64 1" "64 4" "128 4" "256 1" "32 8"; do set -- $v; for i in 1 2; do echo -n "small grid $1 mult $2: "; ESIM_SMALL_GRID=$1 ESIM_SMALL_MULT=$2 timeout -k 10 200 python tools/run_preset.py york | grep us/step | cut -c1-110 || exit 5; done; done
            for v in "0 1" "64 4"; do set -- $v; echo -n "bench20 small grid $1 mult $2: "; ESIM_SMALL_GRID=$1 ESIM_SMALL_MULT=$2 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-extra-runs --cpu-seconds 0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); t=d['config']['timed_region']; print(t['wall_us_per_step']*20, t['chunk_passes_device_ms']*1e3)" || exit 6; done ;;
    *) echo "unknown step $step"; exit 9 ;;
  esac
done
