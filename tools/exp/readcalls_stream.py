#!/usr/bin/env python3
"""Pipeline.ReadCalls with its input in pieces (-B): wall time and peak host memory per piece size, BASELINE config 1
data x scale, plain and gzipped FASTQ.  Run ON the GPU box: python tools/exp/readcalls_stream.py [scale]"""
import gzip
import os
import resource
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kmers_amd import synth  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
kmg = os.path.join(ROOT, "kmers_amd", "kmg")
d = tempfile.mkdtemp(prefix="kmg_rcs_")
taxa, genomes, reads = synth.cfg1(scale)
fa, fq = os.path.join(d, "ref.fa"), os.path.join(d, "reads.fq")
open(fa, "wb").write(synth.fasta_bytes(taxa, genomes))
open(fq, "wb").write(synth.fastq_bytes(reads))
with open(fq, "rb") as f, gzip.open(fq + ".gz", "wb", compresslevel=1) as g:
    while True:
        blk = f.read(1 << 24)
        if not blk:
            break
        g.write(blk)
p = lambda n: os.path.join(d, n)
subprocess.run([kmg, "Database.MakeDatabase", "-a", "-i", fa, "-o", p("ref.db"), "-K", "31", "-k", "31", "-l", "6", "-t", "16"], check=True)
print("reads: %d, FASTQ %.0f MB (gz %.0f MB)" % (reads.shape[0], os.path.getsize(fq) / 1e6, os.path.getsize(fq + ".gz") / 1e6))
first = None
for src in (fq, fq + ".gz"):
    for b in ("1t", "64m", "16m", "4m", "1m"):
        best, rss = 1e9, 0
        for rep in range(2):
            t = time.perf_counter()
            pr = subprocess.Popen([kmg, "Pipeline.ReadCalls", "-q", src, "-d", p("ref.db"), "-o", p("calls"), "-t", "16", "-B", b])
            _, status, ru = os.wait4(pr.pid, 0)
            assert status == 0
            best = min(best, time.perf_counter() - t)
            rss = max(rss, ru.ru_maxrss)
        out = open(p("calls"), "rb").read()
        first = first or out
        assert out == first
        print("%-9s -B %-5s %.3f s   peak RSS %.0f MB" % ("gz" if src.endswith(".gz") else "plain", b, best, rss / 1024))
