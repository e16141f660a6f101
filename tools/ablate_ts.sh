#!/bin/bash
# Builds timing-only variants of gemm_tn_stream (wavenet_autoencoders_amd/libwae_tsabl<tag>.so) next to the product library.
# Arguments: ablation bit masks (-DWAE_TS_ABLATE=<bits>, tag = bits) or "tag:flags", e.g. "k16n6:-DTS_KT=16 -DTS_NS=6".
set -e
cd "$(dirname "$0")/../wavenet_autoencoders_amd/csrc"
make -s
for arg in "$@"; do
  if [[ "$arg" == *:* ]]; then tag="${arg%%:*}"; flags="${arg#*:}"; else tag="$arg"; flags="-DWAE_TS_ABLATE=$arg"; fi
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $flags -c gemm_tn_stream.hip -o /tmp/gemm_ts_abl$tag.o
  objs=$(ls *.o | grep -v '^gemm_tn_stream.o$')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libwae_tsabl$tag.so $objs /tmp/gemm_ts_abl$tag.o
done
