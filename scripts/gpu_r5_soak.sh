#!/bin/bash
# Round 5 closing soak on the final library: scripts/soak_dynamic.py with the phases cycling through Msaa 1/2/4/8 and
# the overlay (42 phases = every combination of the three cycles), then the 5 M liveness stress.
out=gpurun_out/${1:-r5_soak}; mkdir -p $out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $out/build_smoke.log 2>&1; echo "build+smoke rc $?"
SOAK_VERBOSE=1 timeout 300 python scripts/soak_dynamic.py 84000 > $out/soak_dynamic.log 2>&1; echo "soak rc $?"
tail -2 $out/soak_dynamic.log
timeout 240 python scripts/stress_lanes.py > $out/stress_lanes.log 2>&1; echo "stress rc $?"
tail -3 $out/stress_lanes.log
