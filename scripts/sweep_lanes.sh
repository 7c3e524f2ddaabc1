#!/bin/bash
# lanes x streams sweep of the pipelined headline loop (same box)
R=$GRAFT_REPO_ROOT
for cfg in "6 3" "8 4" "4 4" "6 6" "8 8" "5 5" "3 3" "8 2" "4 2"; do
  set -- $cfg
  python $R/scripts/loop_pipelined.py $1 ${FRAMES:-800} ${GS:-1.0} 0 0 $2
done
