import json
import signal
import sys

signal.signal(signal.SIGPIPE, signal.SIG_DFL)   # `| head` is fine
for l in open(sys.argv[1]):
    i = l.find('{"metric')
    if i >= 0:
        d = json.loads(l[i:])
        print("ms/step", round(d["ms_per_step"], 3), "value", round(d["value"] / 1e9, 3), "G/s", "roofline", d["roofline"]["kernel"], d["roofline"]["frac"])
        for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"]):
            print("  %-20s %8.4f ms  x%d  %s GB/s" % (k, v["ms_per_step"], v["launches_per_step"], v["algo_GBps"]))
