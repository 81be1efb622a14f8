#!/bin/bash
# The rocprofv3 passes of one profile set, on the GPU box from the repo root: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE
# in passes of their own (MI355X_MICROARCH.md), then two SQ counter passes.  usage: bash tools/profile_passes.sh tag [preset]
tag=${1:-x}; preset=${2:-uk64m}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$tag; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- python3 tools/run_preset.py $preset 5000 > $out/kt.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/fetch -o f -- python3 tools/run_preset.py $preset 5000 > $out/fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/write -o w -- python3 tools/run_preset.py $preset 5000 > $out/write.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $out/sq -o s -- python3 tools/run_preset.py $preset 5000 > $out/sq.log 2>&1 || exit 1
python3 profiles/summarize_db.py $tag $out/kt/kt_results.db $out/fetch/f_results.db $out/write/w_results.db --workload $preset --steps 5000 > $out/summary.txt 2>&1
cp profiles/${tag}_summary.md profiles/${tag}_summary.json $out/ 2>/dev/null
grep -h "us/step" $out/kt.log | cut -c1-200
tail -n 22 $out/summary.txt
