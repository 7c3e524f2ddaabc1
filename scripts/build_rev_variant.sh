#!/bin/bash
# Build a library VARIANT from the kernel sources of a git REVISION (default HEAD) into gpurun_variants/<name>.so: the
# "before" of a same-box A/B (scripts/ab_variants.sh) while the working tree holds the "after".
#   bash scripts/build_rev_variant.sh <name> [revision] [extra hipcc flags ...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; REV=${2:-HEAD}; shift; shift || true
T=$(mktemp -d)
git -C $R archive $REV bevy_gaussian_splatting_amd/csrc bevy_gaussian_splatting_amd/_build_id.py include | tar -x -C $T
cd $T/bevy_gaussian_splatting_amd/csrc
mkdir -p $R/gpurun_variants
make -s build_id.inc
for f in $(ls *.hip | sed 's/\.hip$//'); do   # (every translation unit of the revision: one file up to round 6's split)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Wall -Wno-unused-function "$@" -c $f.hip -o $T/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/gpurun_variants/$NAME.so $T/*.o
rm -rf $T
echo "built gpurun_variants/$NAME.so (revision $REV $*)"
