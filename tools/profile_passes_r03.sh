#!/bin/bash
# The rocprofv3 passes of one round-3 profile set, on the GPU box from the repo root (every counter group in a pass of its own,
# with --kernel-trace only: MI355X_MICROARCH.md, rocprofv3 PMC slots).  usage: bash tools/profile_passes_r03.sh tag [preset]
#   kt      kernel trace + stats (durations)
#   rd_a    TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum          memory-side read requests, by size ...
#   rd_b    TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum     ... so that read bytes are request-size exact
#   wr      TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum          writes: 64-byte requests and the rest (32 B)
#   at      TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum      atomics executed at the memory side, L2 hit rate
#   fetch / write   FETCH_SIZE, WRITE_SIZE (the derived counters of earlier rounds, for continuity)
#   sq_a    SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES
#   sq_b    SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE
tag=${1:-x}; preset=${2:-uk64m}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$tag; mkdir -p $out
pass() { name=$1; shift; echo "pass $name: $*"; rocprofv3 --kernel-trace "$@" -d $out/$name -o $name -- python3 tools/run_preset.py $preset > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $out/$name.log; return 1; }; }
pass kt --stats || exit 1
pass rd_a --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum || exit 1
pass rd_b --pmc TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum || exit 1
pass wr --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum || exit 1
pass at --pmc TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum || exit 1
pass fetch --pmc FETCH_SIZE || exit 1
pass write --pmc WRITE_SIZE || exit 1
pass sq_a --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES || exit 1
pass sq_b --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE || exit 1
grep -h "us/step" $out/kt.log | cut -c1-220
python3 tools/work_counts.py $tag $preset > $out/work.log 2>&1 || { echo "work counts failed"; tail -5 $out/work.log; }
python3 profiles/summarize_r03.py $tag $out --workload $preset --steps 5000 > $out/summary.txt 2>&1
tail -n 60 $out/summary.txt
