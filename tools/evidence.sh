#!/bin/bash
# run ON the GPU box from the repo root: tools/evidence.sh <tag>  -- the measurements DESIGN.md argues from, on the
# library as built: bench line, rocprofv3 kernel stats of the same command, PMC traffic (two passes, calibrated),
# SQ counter tables (two passes).  Everything lands under gpurun_out/<tag>_*; copy what is to be judged into profiles/.
set -e
TAG=${1:-r2_final}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
# PMC traffic first: the bench line reports roofline.traffic from the newest profiles/*_pmc_traffic.json whose source
# hash is the loaded library's
timeout -k 10 400 python tools/pmc_traffic.py ${TAG}
cp gpurun_out/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json
timeout -k 10 400 python bench.py --steps 20 --warmup 3 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python tools/show_bench.py gpurun_out/${TAG}_bench.json | head -8
rm -rf gpurun_out/prof_${TAG}
(cd /tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/prof_${TAG} -- python3 $OLDPWD/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OLDPWD/gpurun_out/${TAG}_prof_bench.json 2> $OLDPWD/gpurun_out/${TAG}_prof.err)
cp $(ls gpurun_out/prof_${TAG}/*/*kernel_stats.csv | head -1) gpurun_out/${TAG}_kernel_stats.csv
head -12 gpurun_out/${TAG}_kernel_stats.csv
timeout -k 10 300 python tools/pmc_sq.py ${TAG}_a > /dev/null
timeout -k 10 300 python tools/pmc_sq.py ${TAG}_b SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES > /dev/null
echo evidence done
