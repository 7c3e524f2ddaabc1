#!/bin/bash
# A/B of debug-flag variants on the same box: single stream and 6 lanes / 3 streams, dense and scene-like
R=$GRAFT_REPO_ROOT
for gs in ${GSS:-1.0 0.05}; do
for fl in ${FLAGS:-0 0x400000 0x800000}; do
  python $R/scripts/loop_pipelined.py 1 400 $gs 0 $fl
  python $R/scripts/loop_pipelined.py 6 800 $gs 0 $fl 3
done; done
