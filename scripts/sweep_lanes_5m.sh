#!/bin/bash
# lanes x streams on the 5 M f16 frame (dense and scene-like)
R=$GRAFT_REPO_ROOT
export N=5000000 F16=1
for gs in 1.0 0.05; do
for cfg in "1 1" "3 3" "6 3" "4 2" "6 2" "8 4" "6 6" "2 2"; do
  set -- $cfg
  python $R/scripts/loop_pipelined.py $1 300 $gs 0 0 $2
done
done
