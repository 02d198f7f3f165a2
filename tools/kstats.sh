#!/bin/bash
# run ON the GPU box from the repo root: tools/kstats.sh <tag> [bench args]  -- rocprofv3 kernel stats of a short bench run
TAG=${1:-ks}; shift
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -rf gpurun_out/prof_${TAG}
(cd /tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/prof_${TAG} -- python3 $OLDPWD/bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > $OLDPWD/gpurun_out/${TAG}_prof_bench.json 2> $OLDPWD/gpurun_out/${TAG}_prof.err) || exit 1
cp $(ls gpurun_out/prof_${TAG}/*/*kernel_stats.csv | head -1) gpurun_out/${TAG}_kernel_stats.csv
python3 - <<P
import csv
rows=list(csv.DictReader(open('gpurun_out/${TAG}_kernel_stats.csv')))
for r in rows[:45]:
    print("%-64s calls %5s avg %9.1f us  per-step %8.1f us"%(r['Name'][:64], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/37e3))
P
