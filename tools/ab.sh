#!/bin/bash
# A/B on one GPU box: libesim_old.so (a build of an earlier commit) against libesim.so, presets given as arguments, 2 runs each
for p in "$@"; do for i in 1 2; do
  echo -n "old $p "; ESIM_LIB=$PWD/epidemicsimulator_amd/libesim_old.so python3 tools/run_preset.py $p 5000 | grep us/step | cut -d' ' -f5,10 | tr -d ','
  echo -n "new $p "; python3 tools/run_preset.py $p 5000 | grep us/step | cut -d' ' -f5,10 | tr -d ','
done; done
