"""python -m kmers_amd <Database.MakeDatabase | Database.Matcher | Utils.TaxaCounts | Utils.CollectByRead | Utils.ReadClassifier> <options>
-- the reference's option letters for the hot path (README.md:9-12 `java -cp Kmers.jar <Class> <opts>`).
Pipeline.ReadCalls -q reads.fq (-d database | -a reference.fa) -o calls [-K k -l key_length -r read_map.gz]: the chain
MakeDatabase -q | Matcher | CollectByRead | ReadClassifier without its intermediate files (no reference counterpart)."""
import argparse
import sys

from . import Context, programs


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if not argv:
        print(__doc__)
        return 1
    prog, rest = argv[0].split(".")[-1], argv[1:]
    ap = argparse.ArgumentParser(prog=prog, add_help=False)
    if prog == "MakeDatabase":
        for f in "ioKklzrDRPLGm":
            ap.add_argument("-" + f)
        for f in "aqpZvh":
            ap.add_argument("-" + f, action="store_true")
        a = ap.parse_args(rest)
        ctx = Context(0)
        programs.make_database(ctx, a.i, a.o, "a" if a.a else ("p" if a.p else "q"), int(a.K or 32), int(a.k or 24), int(a.l or 6),
                               None if a.Z else int(a.z or 5), a.r, None if a.D is None else int(a.D),
                               None if a.R is None else int(a.R), hr=a.h, passes=int(a.P or 1), limit=a.L,
                               share=None if a.G is None else tuple(int(x) for x in a.G.split("/")), id_map_path=a.m)
    elif prog == "Matcher":
        for f in "idozZnLkK":
            ap.add_argument("-" + f)
        ap.add_argument("-j", action="store_true")
        a = ap.parse_args(rest)
        programs.matcher(Context(0), a.i, a.d, a.o, plain=a.Z is not None, limit=a.L, max_diff=int(a.n or 0), just=a.j,
                         min_k=a.k, max_k=a.K)
    elif prog == "TaxaCounts":
        for f in "do":
            ap.add_argument("-" + f)
        a = ap.parse_args(rest)
        programs.taxa_counts(Context(0), a.d, a.o)
    elif prog in ("CollectByRead", "ReadClassifier"):
        for f in "ioft":
            ap.add_argument("-" + f)
        a = ap.parse_args(rest)
        (programs.collect_by_read if prog == "CollectByRead" else programs.read_classifier)(Context(0), a.i, a.o)
    elif prog == "ReadCalls":
        for f in "qdaoKklr":
            ap.add_argument("-" + f)
        a = ap.parse_args(rest)
        programs.read_calls(Context(0), a.q, a.o, db_path=a.d, fasta_path=a.a, k=int(a.K or 32), key_length=int(a.l or 6),
                            read_map_path=a.r, min_k=None if a.k is None else int(a.k))
    else:
        raise SystemExit("unknown program %s" % argv[0])
    return 0


if __name__ == "__main__":
    sys.exit(main())
