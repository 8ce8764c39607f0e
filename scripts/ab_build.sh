#!/bin/bash
# A/B build of ONE translation unit: scripts/ab_build.sh <name> <file.hip> "<extra flags>" -> ab/lib_<name>.so (the other objects from diffute_amd/build)
# use with DIFFUTE_HIP_LIB=ab/lib_<name>.so (diffute_amd/_cabi.py) - measurement aid; ab/ is git-ignored and travels with gpurun.
# The compile command is the Makefile's own (make -n), per-object flags included (attention.o: -fno-slp-vectorize), so both arms of an A/B share codegen flags.
set -e
name=$1; src=$2; flags=$3
cd "$(dirname "$0")/.."
mkdir -p ab
base=$(basename $src .hip)
cmd=$(make -C diffute_amd/csrc -n -W $src EXTRA="$flags" ../build/${base}.o | grep -- "-c $src" | head -1)
[ -n "$cmd" ] || { echo "ab_build: no compile rule for $src"; exit 1; }
cmd=${cmd/-o ..\/build\/${base}.o/-o ..\/..\/ab\/${base}_${name}.o}
(cd diffute_amd/csrc && eval "$cmd")
objs=$(ls diffute_amd/build/*.o | grep -v "/${base}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o ab/lib_${name}.so $objs ab/${base}_${name}.o
echo built ab/lib_${name}.so
