#!/bin/bash
# run ON the GPU box: the reference's programs with every option at its default (k-mers of 24 .. 32 bases) at config 1/2 size
# (5 Mbp reference, 1 M x 150 bp reads), native driver, wall clock per stage.
set -e
cd "$(dirname "$0")/../.."
D=/tmp/e2e_def; rm -rf $D; mkdir -p $D
python - <<PY
import numpy as np
from kmers_amd import synth
taxa, genomes = synth.random_genomes(1, 10, 500000)
reads = synth.random_reads(2, genomes, 1000000, 150, 0.01)
open("$D/ref.fa","wb").write(synth.fasta_bytes(taxa, genomes))
open("$D/reads.fq","wb").write(synth.fastq_bytes(reads))
PY
t() { local s=$(date +%s%N); "$@"; local e=$(date +%s%N); echo "$(( (e - s) / 1000000 )) ms : ${*:1:3}"; }
t kmers_amd/kmg Database.MakeDatabase -a -i $D/ref.fa -o $D/ref.db
t kmers_amd/kmg Database.MakeDatabase -q -i $D/reads.fq -o $D/reads.db -r $D/reads.map.gz
t kmers_amd/kmg Database.Matcher -i $D/reads.db -d $D/ref.db -o $D/m.gz -v
t kmers_amd/kmg Utils.CollectByRead -i $D/m.gz -o $D/by_read
t kmers_amd/kmg Utils.ReadClassifier -i $D/by_read -o $D/calls
ls -la $D | awk 'NR>3{print $5, $9}'
zcat $D/m.gz | head -2 | cut -c1-150
zcat $D/m.gz | wc -l
