#!/bin/bash
# run ON the GPU box: variable-length build times for every libkmers_gpu*.so variant
cd "$(dirname "$0")/../.."
shopt -s nullglob
for lib in kmers_amd/libkmers_gpu.so kmers_amd/libkmers_gpu_[0-9]*.so; do
  echo "== $lib"; KMERS_GPU_LIB=$PWD/$lib timeout -k 10 200 python tools/exp/var_build_probe.py 2>&1 | tail -1
done
