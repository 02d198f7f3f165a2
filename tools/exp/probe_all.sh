#!/bin/bash
# run ON the GPU box: the extraction probe for the default library and every numbered variant
cd "$(dirname "$0")/../.."
shopt -s nullglob
for rep in 1 2; do
for lib in kmers_amd/libkmers_gpu.so kmers_amd/libkmers_gpu_[0-9]*.so; do
  KMERS_GPU_LIB=$PWD/$lib timeout -k 10 120 python tools/exp/extract_probe.py 2>gpurun_out/probe.err || { echo "$lib FAILED"; tail -3 gpurun_out/probe.err; exit 1; }
done
done
