#!/bin/bash
# The counter calibration (tools/micro/tcc_calib.hip) under the same PMC passes as the profile sets; prints counters per pattern.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/calib; mkdir -p $out
for grp in "rd_a TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "rd_b TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "at TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum" "fetch FETCH_SIZE" "write WRITE_SIZE"; do
  set -- $grp; name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $out/$name -o $name -- ./tools/micro/tcc_calib > $out/$name.log 2>&1 || { echo "calib pass $name failed"; tail -5 $out/$name.log; exit 1; }
done
python3 profiles/summarize_r03.py calib $out --calib > $out/summary.txt 2>&1
cat $out/summary.txt
