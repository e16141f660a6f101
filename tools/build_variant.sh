#!/bin/bash
# Build libwae_<name>.so = the product objects with ONE source rebuilt under extra defines:
#   tools/build_variant.sh <name> <source.hip> "<-Dflags>"      -> wavenet_autoencoders_amd/libwae_<name>.so
set -e
cd "$(dirname "$0")/../wavenet_autoencoders_amd/csrc"
name=$1; src=$2; flags=$3
mkdir -p /tmp/wae_variants
obj=/tmp/wae_variants/${src%.hip}_$name.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-inline-asm $flags -c $src -o $obj
others=$(ls obj/*.o | grep -v "^obj/${src%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libwae_$name.so $obj $others
echo built libwae_$name.so
