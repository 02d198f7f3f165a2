#!/usr/bin/env bash
# Cross-check against the real reference when a JVM exists (none in the build container or on the GPU box).
# Runs config 1 through out/Kmers.jar and through `python -m kmers_amd`, then diffs at tier T1:
# unzipped DB files byte for byte (reference DB), block payloads (read DB, read lists sorted), match lines
# (read lists sorted), TaxaCounts lines and the blocks of CollectByRead / ReadClassifier.  Usage: tests/jar_crosscheck.sh /path/to/Kmers.jar [workdir]
# KMG_DRIVER="kmers_amd/kmg" checks the native C++ drivers instead of the Python ones.
set -euo pipefail
JAR=${1:?path to Kmers.jar}
WORK=${2:-/tmp/kmers_crosscheck}
command -v java >/dev/null || { echo "no java on PATH: parity vs Kmers.jar stays unpinned"; exit 3; }
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$WORK/java_ref" "$WORK/java_reads" "$WORK/gpu" && cd "$WORK"
python3 - <<PY
import sys; sys.path.insert(0, "$ROOT")
from kmers_amd import synth
taxa, genomes, reads = synth.cfg1(0.1)
open("ref.fa", "wb").write(synth.fasta_bytes(taxa, genomes))
open("reads.fq", "wb").write(synth.fastq_bytes(reads))
PY
T=$(nproc)
( cd java_ref   && /usr/bin/time -v java -cp "$JAR" Database.MakeDatabase -a -i ../ref.fa   -o ref.db   -K 31 -k 31 -Z -t "$T" )
( cd java_reads && /usr/bin/time -v java -cp "$JAR" Database.MakeDatabase -q -i ../reads.fq -o reads.db -r reads.map.gz -K 31 -k 31 -Z -t "$T" )
/usr/bin/time -v java -cp "$JAR" Database.Matcher -i java_reads/reads.db -d java_ref/ref.db -o java.matches.txt -Z x -t "$T"
java -cp "$JAR" Utils.TaxaCounts -d java_ref/ref.db -o java.counts.tsv
java -cp "$JAR" Utils.CollectByRead -i java.matches.txt -o java.by_read
java -cp "$JAR" Utils.ReadClassifier -i java.by_read -o java.calls -t "$T"
cd "$ROOT"
DRIVER=${KMG_DRIVER:-python3 -m kmers_amd}
$DRIVER Database.MakeDatabase -a -i "$WORK/ref.fa"   -o "$WORK/gpu/ref.db"   -K 31 -k 31 -Z
$DRIVER Database.MakeDatabase -q -i "$WORK/reads.fq" -o "$WORK/gpu/reads.db" -r "$WORK/gpu/reads.map.gz" -K 31 -k 31 -Z
$DRIVER Database.Matcher -i "$WORK/gpu/reads.db" -d "$WORK/gpu/ref.db" -o "$WORK/gpu.matches.txt" -Z x
$DRIVER Utils.TaxaCounts -d "$WORK/gpu/ref.db" -o "$WORK/gpu.counts.tsv"
$DRIVER Utils.CollectByRead -i "$WORK/gpu.matches.txt" -o "$WORK/gpu.by_read"
$DRIVER Utils.ReadClassifier -i "$WORK/gpu.by_read" -o "$WORK/gpu.calls"
$DRIVER Pipeline.ReadCalls -q "$WORK/reads.fq" -d "$WORK/java_ref/ref.db" -o "$WORK/gpu.calls.fused"      # the Java-built reference, no files in between
cmp "$WORK/java_ref/ref.db" "$WORK/gpu/ref.db" && echo "reference DB: byte-identical"
diff <(zcat "$WORK/java_reads/reads.map.gz") <(zcat "$WORK/gpu/reads.map.gz") && echo "read map: identical"
python3 - <<PY
import sys; sys.path.insert(0, "$ROOT")
from oracle import slow_ref as R
a, b = R.read_indexed_file("$WORK/java_reads/reads.db"), R.read_indexed_file("$WORK/gpu/reads.db")
canon = lambda blocks: {k: [(str(km), sorted(v)) for km, v in R.decode_block(blk, "read")] if k >= 0 else blk for k, blk in blocks.items()}
assert canon(a) == canon(b); print("read DB: payload-identical (read lists sorted)")
def lines(p):
    out = []
    for l in open(p):
        kmer, rp, hit = l.rstrip("\n").split("\t")
        out.append((kmer, sorted(rp.split("|")), hit))
    return out
assert lines("$WORK/java.matches.txt") == lines("$WORK/gpu.matches.txt"); print("match lines: identical (read lists sorted)")
assert open("$WORK/java.counts.tsv").read() == open("$WORK/gpu.counts.tsv").read(); print("TaxaCounts: identical")
assert R.read_indexed_file("$WORK/java.by_read") == R.read_indexed_file("$WORK/gpu.by_read"); print("CollectByRead: identical blocks")
assert R.read_indexed_file("$WORK/java.calls") == R.read_indexed_file("$WORK/gpu.calls"); print("ReadClassifier: identical blocks")
assert R.read_indexed_file("$WORK/java.calls") == R.read_indexed_file("$WORK/gpu.calls.fused"); print("Pipeline.ReadCalls: the same blocks in one go")
PY
