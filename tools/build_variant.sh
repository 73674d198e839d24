#!/bin/bash
# A/B build of libpmhip.so: tools/build_variant.sh <name> <source.hip> <extra hipcc flags...>
# recompiles ONE source with the extra flags, links it with the other objects of the regular build into
# posterior_matching_amd/lib/ab/libpmhip_<name>.so (use with PM_LIB_PATH=...; *.so files are git-ignored)
set -e
cd "$(dirname "$0")/../posterior_matching_amd/csrc"
name=$1; src=$2; shift 2
make -s -j8
mkdir -p ../lib/ab /tmp/pm_ab_$name
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c $src -o /tmp/pm_ab_$name/${src%.hip}.o
objs=""
for o in *.o; do
  if [ "$o" = "${src%.hip}.o" ]; then objs="$objs /tmp/pm_ab_$name/$o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs -o ../lib/ab/libpmhip_$name.so
echo ../lib/ab/libpmhip_$name.so
