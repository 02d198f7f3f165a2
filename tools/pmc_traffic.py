#!/usr/bin/env python3
"""HBM traffic per kernel launch from rocprofv3 PMC counters (run ON the GPU box, from the repo root).

Two separate --pmc passes (FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2: they do not fit one pass) over a
short bench.py run that also launches calib_copy_kernel, whose traffic is known exactly (8-byte-per-lane
streaming copy: 1 GiB read, 1 GiB written).  gfx950's FETCH_SIZE under-reports wide coalesced reads
(MI355X_MICROARCH.md, HBM section), so every kernel's counters are scaled by the factor that makes the
calibration launch come out right for this library's access width.  Output: profiles/<tag>_pmc_traffic.json.
"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r1"
# further arguments go to bench.py instead of the default short config-2 run, e.g.  tools/pmc_traffic.py r3_cfg3 --config 3
# (such a run launches no calibration kernel: the factors are taken from the newest config-2 file under profiles/)
BENCH_ARGS = sys.argv[2:]
CALIB_BYTES = (1 << 27) * 8


def run_pass(counter):
    out = os.path.join(ROOT, "gpurun_out", "pmc_%s_%s" % (TAG, counter))
    # (counters for the library's kernels only: a config-3 run launches tens of thousands of small torch kernels while it
    # makes its input, each of which would be serialised and read out)
    only = ["--kernel-include-regex", "ls_|rs_|extract_|rd_|match_|scan_|number_reads|calib_copy"] if BENCH_ARGS else []
    cmd = ["timeout", "-s", "KILL", "240" if not BENCH_ARGS else "300", "rocprofv3", "--pmc", counter] + only + ["--output-format", "csv", "-d", out, "--",
           sys.executable, os.path.join(ROOT, "bench.py")] + (BENCH_ARGS or ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--calib"])
    env = dict(os.environ, TMPDIR="/tmp")
    if "--config" in BENCH_ARGS and BENCH_ARGS[BENCH_ARGS.index("--config") + 1] == "3":
        env["KMERS_CFG3_KERNELS_ONLY"] = "1"          # one reference build + one chunk: the launches whose bytes are wanted
    subprocess.run(cmd, cwd="/tmp", env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    per = {}
    for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            name = (row["Kernel_Name"], int(row["Grid_Size"]))
            s = per.setdefault(name, [0.0, 0])
            s[0] += float(row["Counter_Value"])
            s[1] += 1
    return per


def main():
    fetch, write = run_pass("FETCH_SIZE"), run_pass("WRITE_SIZE")
    # launches of one kernel with different grids (the read build's and the reference build's) are kept apart
    grids = {}
    for name, g in fetch:
        grids.setdefault(name, set()).add(g)
    label = lambda k: k[0] if len(grids[k[0]]) == 1 else "%s [grid %d]" % k
    fetch = {label(k): v for k, v in fetch.items()}
    write = {label(k): v for k, v in write.items()}
    cals = [k for k in fetch if "calib_copy_kernel" in k]
    if cals:
        cal = cals[0]
        # counters are reported in KiB
        f_fac = CALIB_BYTES / (fetch[cal][0] / fetch[cal][1] * 1024.0)
        w_fac = CALIB_BYTES / (write[cal][0] / write[cal][1] * 1024.0)
    else:
        prev = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json"))
                      if "calib_copy_kernel" in json.load(open(f))["calibration"]["kernel"])
        c = json.load(open(max(prev, key=os.path.getmtime)))["calibration"]
        cal, f_fac, w_fac = "factors of %s" % c["kernel"], c["fetch_factor"], c["write_factor"]
    sys.path.insert(0, ROOT)
    from bench import source_hash
    res = {"source_hash": source_hash(),        # bench.py reports this measurement only for the same kernel sources
           "calibration": {"kernel": cal, "known_bytes_each_way": CALIB_BYTES, "fetch_factor": f_fac, "write_factor": w_fac,
                           "note": "traffic = FETCH_SIZE*1024*fetch_factor + WRITE_SIZE*1024*write_factor, averaged per launch"},
           "kernels": {}}
    for k in sorted(fetch):
        if k not in write:
            continue
        fb = fetch[k][0] / fetch[k][1] * 1024.0 * f_fac
        wb = write[k][0] / write[k][1] * 1024.0 * w_fac
        res["kernels"][k] = {"launches": fetch[k][1], "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
                             "traffic_bytes_per_launch": fb + wb}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", "%s_pmc_traffic.json" % TAG)
    json.dump(res, open(path, "w"), indent=1)
    print("calibration factors: fetch %.3f write %.3f" % (f_fac, w_fac))
    for k, v in sorted(res["kernels"].items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"])[:12]:
        print("%-60s x%-4d %8.1f MB fetched %8.1f MB written" % (k[:60], v["launches"], v["fetch_bytes_per_launch"] / 1e6,
                                                               v["write_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
