#!/bin/bash
# run ON the GPU box: bench every libkmers_gpu*.so variant back to back, twice, print the kernel lines
# AB_GREP: only kernel lines matching this pattern (default: the first AB_LINES lines)
cd "$(dirname "$0")/.."
shopt -s nullglob
for rep in 1 2; do
for lib in kmers_amd/libkmers_gpu.so kmers_amd/libkmers_gpu_[0-9]*.so; do
  KMERS_GPU_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "$lib FAILED"; tail -3 gpurun_out/ab.err; exit 1; }
  echo "== $lib"
  if [ -n "$AB_GREP" ]; then python tools/show_bench.py gpurun_out/ab.json | grep -E "ms/step|$AB_GREP"; else python tools/show_bench.py gpurun_out/ab.json | head -${AB_LINES:-6}; fi
done
done
