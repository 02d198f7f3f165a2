#!/bin/bash
# run ON the GPU box: TaxaCounts time of config 3 at --scale $1 for every libkmers_gpu*.so variant
cd "$(dirname "$0")/../.."
shopt -s nullglob
for lib in kmers_amd/libkmers_gpu.so kmers_amd/libkmers_gpu_[0-9]*.so; do
  KMERS_GPU_LIB=$PWD/$lib timeout -k 10 200 python bench.py --config 3 --scale ${1:-0.3} --no-cpu-baseline > /dev/null 2> gpurun_out/taxa_ab.err || echo "$lib: checks failed (a probe build?)"
  echo "== $lib: $(grep -o 'TaxaCounts: .*' gpurun_out/taxa_ab.err)"
done
