#!/bin/bash
# run ON the GPU box: kernel times of the short-k-mer extraction for every libkmers_gpu*.so variant (rocprofv3 kernel trace)
R="$(cd "$(dirname "$0")/../.." && pwd)"
cd /tmp; export TMPDIR=/tmp
shopt -s nullglob
for lib in $R/kmers_amd/libkmers_gpu.so $R/kmers_amd/libkmers_gpu_[0-9]*.so; do
  export KMERS_GPU_LIB=$lib
  rm -rf /tmp/prof_ab
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab -- python3 $R/tools/exp/var_build_probe.py > /dev/null 2>&1
  echo "== $lib"
  grep -h "extract_short\|il_values\|il_place" /tmp/prof_ab/*/*kernel_stats.csv | awk -F'",' '{split($2,a,","); printf "  %-60s calls %s avg %.3f ms max %.3f ms\n", substr($1,2,60), a[1], a[3]/1e6, a[6]/1e6}'
done
