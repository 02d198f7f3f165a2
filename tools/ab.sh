#!/bin/bash
# A/B timing of kernel variants on ONE box: tools/ab.sh "<-D flags of variant B>" [more variants...]
# builds kmers_amd/libkmers_gpu_<i>.so here (CPU), prints the gpurun command to run them back to back.
set -e
cd "$(dirname "$0")/.."
i=0
for flags in "$@"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-function $flags -o kmers_amd/libkmers_gpu_$i.so kmers_amd/csrc/kmers_gpu.hip
  echo "built variant $i: $flags"
done
