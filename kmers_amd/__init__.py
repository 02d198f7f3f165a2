"""kmers_amd -- MI355X-native k-mer hot path (extract, sort+dedup, match, TaxaCounts) behind a C ABI.
Mirrors the interfaces of danielmoney/kmers for that path only; see DESIGN.md."""
from .api import (Context, DB, FileCreator, KmerDatabase, SequenceBatch, InconsistentDataException,
                  KmersGpuError, match_lines, merge_databases, taxa_counts, taxa_count_lines, kmer_strings,
                  collect_by_read, collect_by_read_blocks, classify_reads, classify_reads_rows, read_call_blocks, match_near, near_match_lines, near_match_lines_var, match_trie)
from .make_database import make_read_db, make_reference_db

__all__ = ["Context", "DB", "FileCreator", "KmerDatabase", "SequenceBatch", "InconsistentDataException",
           "KmersGpuError", "match_lines", "merge_databases", "taxa_counts", "taxa_count_lines", "kmer_strings",
           "collect_by_read", "collect_by_read_blocks", "classify_reads", "classify_reads_rows", "read_call_blocks", "match_near", "near_match_lines", "near_match_lines_var", "match_trie", "make_read_db", "make_reference_db"]
