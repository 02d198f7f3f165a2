"""One rank of the launcher test (tests/test_bench_launch.py): started by bench.launch_ranks with the launch
contract's environment.  Runs the checked step of bench.py's N > 1 line on the CPU (gloo): this rank's read k-mers
against a range-sharded reference, and asserts what bench.py asserts -- the hits summed over ranks equal the hits of
the whole-reference joins.  The CPU oracle stands in for the GPU builds."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    out_dir, behaviour = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "ok"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["LOCAL_RANK"]) == rank
    if behaviour == "fail" and rank == world - 1:
        sys.exit(3)
    import torch
    import torch.distributed as dist
    from kmers_amd import dist as kd, synth
    from oracle.c_oracle import OracleDB
    dist.init_process_group("gloo", rank=rank, world_size=world)
    k, key_len = 15, 3
    taxa, genomes = synth.random_genomes(1, 3, 4000, shared_frac=0.2)
    ref = OracleDB(synth.fasta_bytes(taxa, genomes), 0, k, key_len)             # every rank holds the reference input
    reads = OracleDB(synth.fastq_bytes(synth.random_reads(2 + rank, genomes, 150, 70, 0.01)), 1, k, key_len)
    hist = np.diff(ref.val_off.astype(np.int64)[ref.bucket_off.astype(np.int64)])
    ranges = kd.balanced_key_ranges(hist, world)
    u0, u1 = int(ref.bucket_off[ranges[rank][0]]), int(ref.bucket_off[ranges[rank][1]])
    shard = ref.keys[u0:u1]                                                      # the range this rank owns

    def local_match(q):
        i = np.searchsorted(shard, q)
        ok = (i < len(shard)) & (shard[np.minimum(i, max(len(shard) - 1, 0))] == q) if len(shard) else np.zeros(len(q), bool)
        return np.where(ok, i, -1)

    owner, hit = kd.match_by_key_range(reads.keys, ranges, key_len, local_match)
    whole_hit, whole_n = reads.match(ref)
    assert np.array_equal(hit >= 0, whole_hit >= 0)
    tot = kd.allreduce_sum(np.array([int((hit >= 0).sum()), int(whole_n)]))
    assert tot[0] == tot[1] and tot[0] > 0
    json.dump({"rank": rank, "world": world, "hits_key_range": int(tot[0]), "hits_whole": int(tot[1])},
              open(os.path.join(out_dir, "rank%d.json" % rank), "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
