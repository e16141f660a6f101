#!/bin/bash
# Builds timing-only variants of gemm_tn_static (wavenet_autoencoders_amd/libwae_tsabl<tag>.so, the naming tools/ablate_ts.py loads):
# arguments are ablation bit masks (-DWAE_TQ_ABL=<bits>: 1 no requests, 2 no MFMA / fragment streams) or "tag:flags".
set -e
cd "$(dirname "$0")/../wavenet_autoencoders_amd/csrc"
make -s
for arg in "$@"; do
  if [[ "$arg" == *:* ]]; then tag="${arg%%:*}"; flags="${arg#*:}"; else tag="q$arg"; flags="-DWAE_TQ_ABL=$arg"; fi
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $flags -c gemm_tn_static.hip -o /tmp/gemm_tq_abl$tag.o
  objs=$(ls *.o | grep -v '^gemm_tn_static.o$')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libwae_tsabl$tag.so $objs /tmp/gemm_tq_abl$tag.o
done
