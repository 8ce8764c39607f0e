#!/bin/bash
# A/B build of ONE translation unit: scripts/ab_build.sh <name> <file.hip> "<extra flags>" -> ab/lib_<name>.so (the other objects from diffute_amd/build)
# use with DIFFUTE_HIP_LIB=ab/lib_<name>.so (diffute_amd/_cabi.py) - measurement aid; ab/ is git-ignored and travels with gpurun
set -e
name=$1; src=$2; flags=$3
cd "$(dirname "$0")/.."
mkdir -p ab
base=$(basename $src .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-variable $flags -c diffute_amd/csrc/$src -o ab/${base}_${name}.o
objs=$(ls diffute_amd/build/*.o | grep -v "/${base}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o ab/lib_${name}.so $objs ab/${base}_${name}.o
echo built ab/lib_${name}.so
