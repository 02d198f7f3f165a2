set -u
cd "$(dirname "$0")/../.."
K=kmers_amd/kmg
T=$(mktemp -d)
python - <<PY
import sys; sys.path.insert(0,'.')
from kmers_amd import synth
taxa, genomes = synth.random_genomes(5, 3, 4000)
open("$T/ref.fa","wb").write(synth.fasta_bytes(taxa, genomes))
reads = synth.random_reads(6, genomes, 50, 100, 0.01)
open("$T/r.fq","wb").write(synth.fastq_bytes(reads))
open("$T/empty.fq","wb").close()
open("$T/nl.fq","wb").write(b"\n\n")
open("$T/trunc.fq.gz","wb").write(__import__("gzip").compress(synth.fastq_bytes(reads))[:-20])
PY
run() { timeout -k 5 60 "$@" > $T/out.txt 2> $T/err.txt; echo "rc=$? :: ${*:2:6} :: $(tail -1 $T/err.txt | cut -c1-100)"; }
run $K Database.MakeDatabase -a -i $T/ref.fa -o $T/ref.db -K 31 -k 31 -l 4
run $K Pipeline.ReadCalls -q $T/nonexistent.fq -d $T/ref.db -o $T/c1
run $K Pipeline.ReadCalls -q $T/r.fq -d $T/nonexistent.db -o $T/c2
run $K Pipeline.ReadCalls -q $T/empty.fq -d $T/ref.db -o $T/c3
run $K Pipeline.ReadCalls -q $T/nl.fq -d $T/ref.db -o $T/c4
run $K Pipeline.ReadCalls -q $T/trunc.fq.gz -d $T/ref.db -o $T/c5
run $K Pipeline.ReadCalls -q $T/r.fq -d $T/ref.db -o $T/c6 -B 50
run $K Database.MakeDatabase -q -i $T/nonexistent.fq -o $T/x.db -r $T/x.map -K 31 -k 31 -l 4
run $K Database.MakeDatabase -q -i $T/empty.fq -o $T/e.db -r $T/e.map -K 31 -k 31 -l 4
run $K Database.MakeDatabase -q -i $T/trunc.fq.gz -o $T/t.db -r $T/t.map -K 31 -k 31 -l 4
run $K Database.MakeDatabase -a -i $T/empty.fq -o $T/ea.db -K 31 -k 31 -l 4
run $K Database.MakeDatabase -q -i $T/r.fq -o $T/r.db -r $T/r.map -K 31 -k 31 -l 4
run $K Database.Matcher -i $T/r.db -d $T/ref.db -o $T/m.gz
run $K Database.Matcher -i $T/nonexistent.db -d $T/ref.db -o $T/m2.gz
run $K Database.Matcher -i $T/e.db -d $T/ref.db -o $T/m3.gz
run $K Database.Matcher -i $T/r.db -d $T/r.db -o $T/m4.gz
run $K Database.Matcher -i $T/r.db -d $T/ref.db -o $T/m5.gz -n 2 -j
run $K Utils.TaxaCounts -d $T/ref.db -o $T/tc.tsv
run $K Utils.TaxaCounts -d $T/r.db -o $T/tc2.tsv
run $K Utils.TaxaCounts -d $T/nonexistent.db -o $T/tc3.tsv
run $K Utils.CollectByRead -i $T/m.gz -o $T/cb
run $K Utils.CollectByRead -i $T/nonexistent.gz -o $T/cb2
run $K Utils.CollectByRead -i $T/ref.fa -o $T/cb3
run $K Utils.ReadClassifier -i $T/cb -o $T/calls
run $K Utils.ReadClassifier -i $T/ref.db -o $T/calls2
run $K Utils.ReadClassifier -i $T/nonexistent -o $T/calls3
head -c 300 $T/ref.db > $T/cut.db
run $K Utils.TaxaCounts -d $T/cut.db -o $T/tc4.tsv
run $K Database.Matcher -i $T/r.db -d $T/cut.db -o $T/m6.gz
rm -rf $T
