#!/bin/bash
# rocprofv3 per-kernel averages of blocking bgs_sort with every splat drawable (SortMode::Rayon keys), 1 M and 5 M, the
# digit passes (flags 0x80000) against the bucket path: one process per combination so that a kernel's average is one workload's.
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5_sort_prof}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for n in 1000000 5000000; do for flags in 0x80000 0x0; do
  tag=${n}_rayon_$flags
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp_$tag -o sp -- python $GRAFT_REPO_ROOT/scripts/sort_rates.py $n rayon $flags > $out/$tag.txt 2>$out/$tag.err
  f=$(find /tmp/sp_$tag -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $out/${tag}_kernel_stats.csv
  cat $out/$tag.txt; head -8 $out/${tag}_kernel_stats.csv | cut -c1-200
done; done
