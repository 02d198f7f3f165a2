#!/usr/bin/env python3
"""End-to-end (files in, files out) timing of the native drivers on BASELINE config 1 data, next to the
device-resident kernel numbers of bench.py (report both).  Run ON the GPU box:
    python tools/e2e_bench.py [scale] > gpurun_out/e2e.json
Writes its inputs under $TMPDIR; every stage is one `kmers_amd/kmg` process (so each includes context
creation, file I/O, host parsing, PCIe copies, gzip)."""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kmers_amd import synth  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
kmg = os.path.join(ROOT, "kmers_amd", "kmg")
d = tempfile.mkdtemp(prefix="kmg_e2e_")
taxa, genomes, reads = synth.cfg1(scale)
fa, fq = os.path.join(d, "ref.fa"), os.path.join(d, "reads.fq")
open(fa, "wb").write(synth.fasta_bytes(taxa, genomes))
open(fq, "wb").write(synth.fastq_bytes(reads))
n_ref_win = sum(len(g) - 30 for g in genomes)
n_read_win = reads.shape[0] * (reads.shape[1] - 30)


def run(*args):
    t = time.perf_counter()
    subprocess.run([kmg] + list(args), check=True)
    return time.perf_counter() - t


p = lambda n: os.path.join(d, n)
res = {"config": {"workload": "BASELINE config 1 x %g, k=31, l=6, files on local disk, one kmg process per stage" % scale,
                  "ref_windows": n_ref_win, "read_windows": n_read_win, "threads": 16}}
common = ["-K", "31", "-k", "31", "-l", "6", "-t", "16"]
res["make_ref_z5_s"] = run("Database.MakeDatabase", "-a", "-i", fa, "-o", p("ref.db"), *common)
res["make_reads_z5_s"] = run("Database.MakeDatabase", "-q", "-i", fq, "-o", p("reads.db"), "-r", p("map.gz"), *common)
res["make_reads_unzipped_s"] = run("Database.MakeDatabase", "-q", "-i", fq, "-o", p("readsZ.db"), "-r", p("mapZ.gz"), "-Z", *common)
res["matcher_s"] = run("Database.Matcher", "-i", p("reads.db"), "-d", p("ref.db"), "-o", p("matches.gz"), "-t", "16", "-v")
res["matcher_n1_s"] = run("Database.Matcher", "-i", p("reads.db"), "-d", p("ref.db"), "-o", p("matches.n1.gz"), "-t", "16", "-n", "1", "-v")
res["taxacounts_s"] = run("Utils.TaxaCounts", "-d", p("ref.db"), "-o", p("counts.tsv"))
res["collect_by_read_s"] = run("Utils.CollectByRead", "-i", p("matches.gz"), "-o", p("by_read"), "-v")
res["read_classifier_s"] = run("Utils.ReadClassifier", "-i", p("by_read"), "-o", p("calls"), "-t", "16", "-v")
# the same calls file from the reads and the reference database in one process, nothing in between on disk
res["read_calls_fused_s"] = run("Pipeline.ReadCalls", "-q", fq, "-d", p("ref.db"), "-o", p("calls.fused"), "-t", "16", "-v")
res["read_calls_fused_from_fasta_s"] = run("Pipeline.ReadCalls", "-q", fq, "-a", fa, "-o", p("calls.fused2"), "-K", "31", "-l", "6", "-t", "16")
res["fused_equals_staged"] = open(p("calls.fused"), "rb").read() == open(p("calls"), "rb").read() == open(p("calls.fused2"), "rb").read()
res["staged_chain_s"] = res["make_reads_z5_s"] + res["matcher_s"] + res["collect_by_read_s"] + res["read_classifier_s"]
res["sizes_MB"] = {n: round(os.path.getsize(p(n)) / 1e6, 1) for n in ("ref.db", "reads.db", "readsZ.db", "matches.gz", "by_read", "calls")}
res["built_kmers_per_s_end_to_end"] = (n_ref_win + n_read_win) / (res["make_ref_z5_s"] + res["make_reads_z5_s"])
print(json.dumps(res))
subprocess.run(["rm", "-rf", d])
