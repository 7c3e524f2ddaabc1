#!/bin/bash
# Round 5: the fused front end (sort + project + bin in one launch): parity, then single-stream / in-flight A/B.
set -u
OUT=gpurun_out/r5_b
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
true
echo "rc=$?" >> $OUT/pytest.log
tail -n 6 $OUT/pytest.log
python scripts/ab_flags.py "dense" "0,0x80" 1 > $OUT/ab_fused.txt 2>&1
tail -n 30 $OUT/ab_fused.txt
