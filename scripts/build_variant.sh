#!/bin/bash
# Build a variant of the kernel library beside the shipped one, for same-box A/B through NSAMD_LIB:
#   scripts/build_variant.sh <name> "<extra hipcc flags>" [patch ...]   ->  nerfstudio_amd/libnsamd_<name>.so
# The sources are copied to a scratch directory first (patches from nerfstudio_amd/csrc/experiments/ are applied to the copy),
# so the tree and the shipped library stay as they are.
set -e
name=$1; flags=$2; shift 2 || true
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d /tmp/nsamd_variant_XXXX)
mkdir -p $tmp/nerfstudio_amd $tmp/include
cp -r $root/nerfstudio_amd/csrc $tmp/nerfstudio_amd/csrc
cp $root/include/nsamd.h $tmp/include/
rm -rf $tmp/nerfstudio_amd/csrc/build
for p in "$@"; do (cd $tmp && patch -p1 < $root/nerfstudio_amd/csrc/experiments/$p); done
make -C $tmp/nerfstudio_amd/csrc -j8 FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics -fPIC -Wall -Wno-unused-function $flags" > $tmp/build.log 2>&1 || { tail -30 $tmp/build.log; exit 1; }
cp $tmp/nerfstudio_amd/libnsamd.so $root/nerfstudio_amd/libnsamd_$name.so
echo "built nerfstudio_amd/libnsamd_$name.so"
rm -rf $tmp
