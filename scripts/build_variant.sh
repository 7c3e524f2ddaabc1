#!/bin/bash
# Build a library VARIANT for same-box A/B (scripts/ab_variants.sh): the tree's kernel sources with extra compiler flags
# (-D switches of experiment code paths, e.g. -DBGS_ABLATION=1 for the kernel-ablation bits of scripts/ablate.py) into
# gpurun_variants/<name>.so. Variants are loaded through BGS_LIB_OVERRIDE only (bench.py and the tests refuse that).
#   bash scripts/build_variant.sh <name> [extra hipcc flags ...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
T=$(mktemp -d)
mkdir -p $R/gpurun_variants
cd $R/bevy_gaussian_splatting_amd/csrc
make -s build_id.inc
for f in $(ls *.hip | sed 's/\.hip$//'); do   # (every translation unit of the revision: one file up to round 6's split)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Wall -Wno-unused-function "$@" -c $f.hip -o $T/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/gpurun_variants/$NAME.so $T/*.o
rm -rf $T
echo "built gpurun_variants/$NAME.so ($*)"
