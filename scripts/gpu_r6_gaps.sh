#!/bin/bash
# launch-boundary gaps of blocking-ish frames (one lane) under several switches: bash scripts/gpu_r6_gaps.sh
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for cfg in "0" "0x4000000" "graphs"; do
  rm -rf /tmp/gp
  if [ "$cfg" = "graphs" ]; then export BGS_LOOP_GRAPHS=1; fl=0; else unset BGS_LOOP_GRAPHS; fl=$cfg; fi
  rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -o t -- python $R/scripts/loop_pipelined.py 1 60 1.0 0 $fl 1 > /tmp/gp.out 2>/tmp/gp.err
  echo "== flags/config $cfg: $(cat /tmp/gp.out)"
  for f in $(find /tmp/gp -name "*kernel_trace.csv"); do python $R/scripts/boundary_gaps.py $f 60; done
done
