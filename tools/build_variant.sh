#!/bin/bash
# Kernel-development variants of the library: tools/build_variant.sh <tag> "<-D flags>" <src.hip> [more.hip ...]
# recompiles only the named sources with the extra flags, reuses the release objects of the others and links
# ssl4gie_amd/libssl4gie_hip_x<tag>.so (load with SSL4GIE_DEBUG_LIB=x<tag>).  Never shipped: experiments only.
set -eu
tag=$1; defs=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$root/ssl4gie_amd/csrc
make -C $csrc -j8 > /dev/null
bdir=$csrc/build_x$tag; mkdir -p $bdir
objs=""
for f in $(ls $csrc/*.hip); do
  b=$(basename $f .hip)
  if [[ " $* " == *" $b.hip "* ]]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$csrc -I$root/include -Wall -Wno-unused-function -ffp-contract=fast $defs -c $f -o $bdir/$b.o &
    objs="$objs $bdir/$b.o"
  else
    objs="$objs $csrc/build/$b.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/ssl4gie_amd/libssl4gie_hip_x$tag.so $objs
echo "built libssl4gie_hip_x$tag.so"
