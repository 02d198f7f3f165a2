#!/usr/bin/env python3
"""Per-kernel SQ counters (instruction mix, LDS conflicts, wait cycles) from one rocprofv3 --pmc pass over
a short bench.py run.  Run ON the GPU box from the repo root:  python tools/pmc_sq.py <tag> [counters...]
At most 8 SQ counters fit one pass.  Output: gpurun_out/<tag>_pmc_sq.txt (per launch averages)."""
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r1"
COUNTERS = sys.argv[2:] or ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES",
                            "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT"]


def main():
    assert len(COUNTERS) <= 8
    out = os.path.join(ROOT, "gpurun_out", "pmcsq_%s" % TAG)
    cmd = ["timeout", "-s", "KILL", "240", "rocprofv3", "--pmc"] + COUNTERS + ["--output-format", "csv", "-d", out, "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    per = {}
    for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = per.setdefault(row["Kernel_Name"], {})
            c = k.setdefault(row["Counter_Name"], [0.0, 0])
            c[0] += float(row["Counter_Value"])
            c[1] += 1
    lines = ["%-44s %s" % ("kernel", " ".join("%14s" % c[3:17] for c in COUNTERS))]
    rows = sorted(per.items(), key=lambda kv: -kv[1].get(COUNTERS[0], [0, 1])[0])
    for name, k in rows[:24]:
        vals = [k.get(c, [0, 1]) for c in COUNTERS]
        lines.append("%-44s %s" % (name[:44], " ".join("%14.4g" % (v[0] / max(v[1], 1)) for v in vals)))
    txt = "\n".join(lines)
    open(os.path.join(ROOT, "gpurun_out", "%s_pmc_sq.txt" % TAG), "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
