#!/bin/bash
# Build a library VARIANT from the WORKING TREE's kernel sources with extra hipcc flags (experiment macros) into
# gpurun_variants/<name>.so (scripts/build_rev_variant.sh does the same from a git revision).
#   bash scripts/build_tree_variant.sh <name> [extra hipcc flags ...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
T=$(mktemp -d)
mkdir -p $T/bevy_gaussian_splatting_amd $T/include
cp -r $R/bevy_gaussian_splatting_amd/csrc $T/bevy_gaussian_splatting_amd/csrc
cp $R/bevy_gaussian_splatting_amd/_build_id.py $T/bevy_gaussian_splatting_amd/
cp $R/include/*.h $T/include/
cd $T/bevy_gaussian_splatting_amd/csrc
rm -f *.o libbgs.so
mkdir -p $R/gpurun_variants
make -s build_id.inc
for f in $(ls *.hip | sed 's/\.hip$//'); do   # (every translation unit of the revision: one file up to round 6's split)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Wall -Wno-unused-function "$@" -c $f.hip -o $T/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/gpurun_variants/$NAME.so $T/*.o
rm -rf $T
echo "built gpurun_variants/$NAME.so (working tree $*)"
