#!/bin/bash
# Build a library VARIANT from the tree's kernel sources with a PATCH applied (scripts/experiments/*.patch), into
# gpurun_variants/<name>.so, for same-box A/B through scripts/ab_variants.sh. The tree's own sources stay untouched (and
# with them bgs_build_id and the stamp of the committed counter files).
#   bash scripts/build_patched_variant.sh <name> <patch file> [extra hipcc flags ...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; PATCH=$(cd "$(dirname "$2")" && pwd)/$(basename "$2"); shift; shift
T=$(mktemp -d)
mkdir -p $T/bevy_gaussian_splatting_amd
cp -r $R/bevy_gaussian_splatting_amd/csrc $T/bevy_gaussian_splatting_amd/csrc
cp $R/bevy_gaussian_splatting_amd/_build_id.py $T/bevy_gaussian_splatting_amd/   # (the Makefile's build-id rule looks one directory up)
cp -r $R/include $T/include                                                        # (bgs_api.hip includes ../../include/bgs.h)
cd $T/bevy_gaussian_splatting_amd/csrc
rm -f *.o libbgs.so build_id.inc
patch -p1 < $PATCH
mkdir -p $R/gpurun_variants
make -s build_id.inc
for f in $(ls *.hip | sed 's/\.hip$//'); do   # (every translation unit of the revision: one file up to round 6's split)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Wall -Wno-unused-function "$@" -c $f.hip -o $T/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/gpurun_variants/$NAME.so $T/*.o
rm -rf $T
echo "built gpurun_variants/$NAME.so (patch $(basename $PATCH) $*)"
