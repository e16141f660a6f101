#!/bin/bash
# PMC passes over wae_glu_layer_fwd (C2 shape).  Usage: tools/pmc_glu.sh <nw 4|8> [out_dir]   (run on the GPU box)
# Thin wrapper over tools/pmc_run.sh (separate rocprofv3 run per counter set, each under `timeout`; no TA_* counters:
# a TA pass crashed rocprofv3 and held the box until gpurun's limit).
NW=${1:-4}
OUT=${2:-gpurun_out/pmc_glu_nw$NW}
exec "$(dirname "$0")/pmc_run.sh" "$OUT" glu_fwd tools/run_glu.py "$NW" 0 0 6
