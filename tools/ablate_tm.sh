#!/bin/bash
# Builds timing-only ablations of gemm_tm (wavenet_autoencoders_amd/libwae_tmabl<bits>.so) next to the product library.
set -e
cd "$(dirname "$0")/../wavenet_autoencoders_amd/csrc"
make -s
for bits in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DWAE_TM_ABLATE=$bits -c gemm_tm.hip -o /tmp/gemm_tm_abl$bits.o
  objs=$(ls *.o | grep -v '^gemm_tm.o$')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libwae_tmabl$bits.so $objs /tmp/gemm_tm_abl$bits.o
done
