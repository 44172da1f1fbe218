#!/bin/bash
# build igmc_amd/lib/libigmc_hip_<name>.so from the kernel sources of a git ref (same-box A/B baselines: tools/gpu_ab_new.sh)
#   tools/build_variant.sh <name> [git-ref = HEAD]
set -e
cd "$(dirname "$0")/.."
name=$1; ref=${2:-HEAD}
d=.scratch/variants/$name
rm -rf $d && mkdir -p $d/x/y/csrc
git archive $ref igmc_amd/csrc include | tar -x -C $d
mv $d/igmc_amd/csrc/* $d/x/y/csrc/ && mv $d/include $d/x/include
IGMC_CSRC_DIR=$PWD/$d/x/y/csrc IGMC_HIP_LIB_OUT=$PWD/igmc_amd/lib/libigmc_hip_$name.so python igmc_amd/build.py --force
