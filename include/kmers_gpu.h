/* kmers_gpu.h -- C ABI of the MI355X-native k-mer hot path (libkmers_gpu.so).
 *
 * Drop-in boundary for three call sites of danielmoney/kmers (paths relative to the reference's
 * src/; the reference has no FFI of its own, these are the Java seams a JNI/FFM shim replaces --
 * see INTEGRATION.md):
 *   build  : KmerFiles/FileCreator.java:33,65,117,219  (new FileCreator / addKmers / create / close)
 *            fed by OtherFiles/KmersFromFile.java:28-32 (streamFromFile)
 *   match  : Database/DB.java:76-90,129-137             (getNearestKmers quick branch)
 *   counts : Utils/TaxaCounts.java:355-360,386-395      (SingleIndexAll.call / CountDataAll.add)
 *
 * Conventions: plain pointers and sizes only.  Every function returns 0 (KMG_OK) or a negative
 * code and never throws/aborts across the ABI; kmg_last_error() gives the message for the last
 * nonzero return on that context.  The caller allocates and frees every host buffer; the library
 * copies inputs before returning and retains no host pointer.  A context is bound to one GPU and is
 * internally serialised (safe to call from several threads; calls do not overlap on the device).
 * Supported: 1 <= minK <= maxK = k <= 32, 1 <= key_len <= min(minK, 12); the exact join takes k-mers of one
 * length (restrict variable-length databases first, kmg_db_restrict).  k-mers of up to 64 bases ("two-word
 * k-mers", see below; 32 < maxK <= 64, any minK): build, fetch, load, encode, restrict, the exact join, the search over several
 * lengths, the inexact search, TaxaCounts and the per-read tools, the merge of file sets, the RCCL steps (every entry point but
 * kmg_match_keys_dev, which has kmg_match_keys_wide_dev beside it).  Anything else returns KMG_EUNSUPPORTED so a shim can fall back to the Java path.
 *
 * K-mer representation everywhere: uint64, 2 bits per base, code order A=0 C=1 T=2 G=3
 * (Kmers/Base.java:7-10), first base in the top bits, low 64-2k bits zero -- i.e. the bytes of
 * Sequence.cBytes() (Kmers/Sequence.java:90-115) read big-endian.  Unsigned integer order of this
 * value == Sequence.compareTo order (Kmers/Sequence.java:165-187) for fixed k.
 * k > 32 (the reference's Sequence is a byte array of any length, Sequence.java:15-35,90-115): a k-mer is TWO such
 * words -- `keys` holds bases 0 .. 31 as above, `keys_lo` bases 32 .. k-1, first of them in the top bits, the low
 * 128-2k bits zero; the two words, big-endian one after the other and cut to ceil(k/4) bytes, are cBytes(), and the
 * order of the (keys, keys_lo) pairs is Sequence.compareTo's.  Kmer.key(key_len) (Kmer.java:47-55) lies in `keys`.
 */
#ifndef KMERS_GPU_H
#define KMERS_GPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KMG_OK            0
#define KMG_EINVAL       (-1)   /* bad argument / inconsistent parameters (InconsistentDataException) */
#define KMG_ENOMEM       (-2)   /* host or device allocation failed */
#define KMG_EDEVICE      (-3)   /* HIP runtime error */
#define KMG_EUNSUPPORTED (-4)   /* k > 64, a join over several k-mer lengths, > 2^32-16 records in one build ... */
#define KMG_ESTATE       (-5)   /* call out of order (add before begin, ...) */

#define KMG_PAYLOAD_TAXON   0   /* reference DB: int32 taxon per sequence  (DataCollector.getCountInstance) */
#define KMG_PAYLOAD_READPOS 1   /* read DB: (int32 read, int16 pos)        (DataCollector.getReadInstance)  */

typedef struct kmg_ctx kmg_ctx;   /* one per (process, GPU) */
typedef struct kmg_db  kmg_db;    /* a device-resident sorted, deduplicated k-mer database (CSR) */

/* ---- context -------------------------------------------------------------------------------- */
int  kmg_create(int device, kmg_ctx **out);
void kmg_destroy(kmg_ctx *ctx);
const char *kmg_last_error(kmg_ctx *ctx);
/* The HIP stream all of this context's kernels run on (a hipStream_t), for callers that time
 * or order work around the library. */
void *kmg_stream(kmg_ctx *ctx);

/* ---- stage 1+2: build  (replaces FileCreator.java:33 ctor .. :117 create) -------------------
 * kmg_build_begin : new FileCreator(temp, keyLength, maxK, cache, collector, rc, ...)
 *                   (FileCreator.java:33-63).  rc != 0 adds the reverse complement of every k-mer
 *                   that is not a palindrome (FileCreator.java:83-87, Kmer.java:210-225).
 * kmg_build_add   : KmersFromFile.streamFromFile + FileCreator.addKmers (KmersFromFile.java:84-208,
 *                   FileCreator.java:65-110) for a batch of nseq sequences.  The host keeps the
 *                   ID/KMER/OTHER record splitting; `bases` holds, back to back, exactly the
 *                   characters > 32 of each sequence (every one of them advances `pos`,
 *                   KmersFromFile.java:154-158); sequence i is bases[seq_off[i] .. seq_off[i+1]).
 *                   Every window of k letters from ACGTacgt is a k-mer; a window holding any
 *                   other character is skipped and the window keeps rolling (:161-177).
 *                   KMG_PAYLOAD_TAXON  : seq_id[i] = taxon of sequence i; seq_tok unused (NULL).
 *                   KMG_PAYLOAD_READPOS: seq_tok[i] = any int64 that is equal for two sequences
 *                     iff their header strings are equal; seq_id unused (NULL).  The library
 *                     numbers reads as ReadIDMapping.geNext does (ReadIDMapping.java:14-23): a
 *                     read gets the next index at its first valid k-mer unless its header equals
 *                     that of the last read that produced a k-mer.  read_index_out (nseq, may be
 *                     NULL) receives each sequence's index, or -1 if it produced no k-mer, so the
 *                     host can write the read-name map (MakeDatabase.java:165-185).  Position is
 *                     the Java (short)(pos - k) (KmersFromFile.java:165).
 *                   KMG_PAYLOAD_TAXON with read_index_out given: it receives 0 for every sequence that
 *                     produced a k-mer and -1 for the others.  The reference turns a header into its
 *                     taxon only when a sequence hands out its first k-mer (new Kmer(..) is evaluated
 *                     before mapper.apply, KmersFromFile.java:103,165), so a header that is no int (or
 *                     not in the -m map) fails there only for the sequences flagged 0: the host passes
 *                     any taxon for such a header and throws if the flag comes back 0.
 *                   May be called many times; batches are consumed in call order.
 * kmg_build_add_dev: the same with every pointer a DEVICE pointer on this context's GPU (inputs
 *                   already resident in HBM); asynchronous on kmg_stream(); the buffers must stay
 *                   valid until kmg_build_finish returns.  read_index_out is a device pointer.
 *                   The offsets are the caller's to get right here (kmg_build_add checks them on the host,
 *                   KMG_EINVAL): seq_off[0] = 0, non-decreasing, seq_off[nseq] = nbases.
 * kmg_build_finish: FileCreator.create's compute (MakeAndWriteKey.call, FileCreator.java:271-347):
 *                   sort by k-mer, deduplicate, reduce.  Reference DB: per k-mer the ascending
 *                   taxon list with occurrence counts (TreeCountMap; counts are NOT saturated
 *                   here, CountDataType.compress saturates at 255 when writing).  Read DB: per
 *                   k-mer every (read, pos) ascending (the reference's HashSet order is
 *                   nondeterministic; ascending is the canonical form).  *out owns device memory;
 *                   release with kmg_db_free.  n_windows (may be NULL) = valid forward windows.
 */
int kmg_build_begin(kmg_ctx *ctx, int k, int key_len, int rc, int payload_kind);
/* kmg_build_set_filters: MakeDatabase -D <threshold> / -R <length> (MakeDatabase.java:279-291): the
 *                   per-k-mer predicates Kmers/Dust.java:12-30 (keep iff sum c(c-1) over the triplet
 *                   counts < threshold*k*2) and Kmers/RunOfSame.java:12-36 (drop iff a run of equal
 *                   bases reaches length), applied to the forward k-mer before FileCreator.addKmers
 *                   (its RC goes with it); reads are numbered before the filters.  -1 = off.  Call
 *                   between kmg_build_begin and the first kmg_build_add. */
int kmg_build_set_filters(kmg_ctx *ctx, int dust_threshold, int runofsame_length);
/* kmg_build_set_min_k: MakeDatabase -k <minK> below -K <maxK> = the k of kmg_build_begin (the reference's defaults are
 *   -k 24 -K 32, MakeDatabase.java:106-107).  Replaces the `ending` branch of KmersFromFileSpliterator.tryAdvance
 *   (OtherFiles/KmersFromFile.java:90-111): besides every full window the build then holds the suffixes of length
 *   min(pos, maxK - 1) .. minK that end at a sequence's last base (:128-133) and those that end at the last base of a
 *   full window with an invalid base in it (:167-176), each skipped on its own if it holds an invalid base; when the
 *   last full window of a sequence is invalid both rules fire and those suffixes are recorded twice, as there.  A read
 *   is numbered at its first k-mer of any length.  The database is ordered as Sequence.compareTo orders k-mers
 *   (Kmers/Sequence.java:165-187): zero-padded value first, length second (kmg_db_fetch_lens).  key_len <= min_k
 *   (Kmer.key of a shorter k-mer throws in the reference).  After kmg_build_begin, before the first add.  Any k <= 64:
 *   beyond 32 the k-mers of up to 32 bases have a zero second word (kmg_db_fetch_keys_lo). */
int kmg_build_set_min_k(kmg_ctx *ctx, int min_k);
/* kmg_build_set_key_range: build only the part of the database whose key buckets (Kmer.key(key_len), Kmer.java:47-55)
 *   lie in [first_bucket, end_bucket) -- each record is judged by its own k-mer, so a reverse complement can be kept
 *   where its forward k-mer is not.  Read numbering is not affected (as with the -D / -R filters: the mapper runs
 *   first).  The reference builds its files key by key as well (FileCreator.MakeAndWriteKey, FileCreator.java:271-406),
 *   so the blocks of several range builds over the same input, written in key order, are the file of one complete
 *   build: inputs whose records do not fit one build (2^32-16) or the GPU's memory are built in passes, and ranks of
 *   a multi-GPU build extract only what they own (first == end: an empty share, an empty database).  After
 *   kmg_build_begin, before the first add.  *n_windows of
 *   kmg_build_finish then counts the forward windows kept. */
int kmg_build_set_key_range(kmg_ctx *ctx, uint32_t first_bucket, uint32_t end_bucket);
int kmg_build_add(kmg_ctx *ctx, const uint8_t *bases, uint64_t nbases, const uint64_t *seq_off,
                  const int32_t *seq_id, const int64_t *seq_tok, uint64_t nseq,
                  int32_t *read_index_out);
int kmg_build_add_dev(kmg_ctx *ctx, const uint8_t *d_bases, uint64_t nbases, const uint64_t *d_seq_off,
                      const int32_t *d_seq_id, const int64_t *d_seq_tok, uint64_t nseq,
                      int32_t *d_read_index_out);
/* kmg_build_add_packed: the same for 2-bit packed sequences as MakeDatabase -p reads them (MakeDatabase.java:320-420,
 *   SequenceDataType.java:21-34): `packed` = Sequence.cBytes of every sequence back to back (first base in the top
 *   bits of a byte, every sequence on a byte boundary), byte_off[nseq+1] its byte offsets, seq_off[nseq+1] the base
 *   offsets.  A quarter of the bytes cross PCIe; the letters are restored on the device.  Taxon payload only. */
int kmg_build_add_packed(kmg_ctx *ctx, const uint8_t *packed, uint64_t nbytes, const uint64_t *seq_off,
                         const uint64_t *byte_off, const int32_t *seq_id, uint64_t nseq);
int kmg_build_finish(kmg_ctx *ctx, kmg_db **out, uint64_t *n_windows);

/* ---- database handles ----------------------------------------------------------------------- */
/* kmg_db_info : sizes.  n_unique = distinct k-mers, n_values = total list entries
 *               ((k-mer, taxon) pairs | (read, pos) occurrences). */
int kmg_db_info(const kmg_db *db, int *k, int *key_len, int *rc, int *payload_kind,
                uint64_t *n_unique, uint64_t *n_values);
/* Width of the packed (read, pos) values a read database built here holds on the device: 32 when the build could
 * keep them in 32 bits (one batch, pos < 2^b from its longest read, fewer than 2^(32-b) reads), else 64; 0 for
 * taxon databases and loaded ones.  Introspection for tests and sizing -- no counterpart in the reference. */
int kmg_db_payload_bits(const kmg_db *db);
/* kmg_db_fetch: copy the CSR to host buffers (any may be NULL to skip):
 *   keys[n_unique] ascending; val_off[n_unique+1]; val_a[n_values] = taxon | read;
 *   val_b[n_values] = count | pos; bucket_off[4^key_len + 1] = first k-mer index of each
 *   Kmer.key(key_len) bucket (Kmer.java:47-55) -- the per-key blocks of the DB file. */
int kmg_db_fetch(const kmg_db *db, uint64_t *keys, uint64_t *val_off, int32_t *val_a, int32_t *val_b,
                 uint64_t *bucket_off);
/* Two-word k-mers (k > 32; Kmers/Sequence.java:90-115 packs any length, Kmer.java:90-97 writes it).
 *   kmg_db_fetch_keys_lo: keys_lo[n_unique] = bases 32 .. k-1 of every k-mer (zeros for a database of k <= 32); the
 *     first 32 bases come with kmg_db_fetch's keys.
 *   kmg_db_load_wide: kmg_db_load / kmg_db_load_var for 32 < k <= 64 -- what KmerFile's decoder (KmerFile.java:200-250)
 *     yields for such a file; min_k = k and lens NULL for k-mers of one length, else lens[n_unique] as in
 *     kmg_db_load_var; (keys, keys_lo, lens) strictly ascending (Sequence.compareTo: zero-padded value, then length), bits
 *     behind a k-mer's last base zero. */
int kmg_db_fetch_keys_lo(const kmg_db *db, uint64_t *keys_lo);
int kmg_db_load_wide(kmg_ctx *ctx, int k, int min_k, int key_len, int rc, int payload_kind, const uint64_t *keys,
                     const uint64_t *keys_lo, const uint8_t *lens, uint64_t n_unique, const uint64_t *val_off,
                     const int32_t *val_a, const int32_t *val_b, kmg_db **out);
/* Variable-length databases (minK < maxK).  kmg_db_min_k: the shortest k-mer length (== k of kmg_db_info for a
 *   database of one length).  kmg_db_fetch_lens: lens[n_unique] = number of bases of every k-mer (the length byte of
 *   Kmer.compressedBytes, Kmer.java:90-97); keys are zero-padded behind it.  kmg_db_load_var: kmg_db_load with lens
 *   (what KmerFile's decoder, KmerFile.java:200-250, yields for such a file); (keys, lens) strictly ascending. */
int kmg_db_min_k(const kmg_db *db);
int kmg_db_fetch_lens(const kmg_db *db, uint8_t *lens);
int kmg_db_load_var(kmg_ctx *ctx, int k, int min_k, int key_len, int rc, int payload_kind, const uint64_t *keys,
                    const uint8_t *lens, uint64_t n_unique, const uint64_t *val_off, const int32_t *val_a,
                    const int32_t *val_b, kmg_db **out);
/* kmg_db_restrict: KmerUtils.restrictedStream(stream, length, length, dataType) (Kmers/KmerUtils.java:59-74) over a whole
 *   database: k-mers shorter than `length` are dropped, the others cut to `length` bases (Kmer.limitTo, Kmer.java:277-289)
 *   and equal neighbours folded with the data type's reducer (counts add per taxon, read-position sets unite).  That is
 *   what Matcher -k L -K L does to the search side (KmerFile.allRestrictedKmers, Matcher.java:241) and to the database
 *   side (DB.java:134) before the exact join.  min_k <= length <= k of the source (DB.java:78-81 -> KMG_EINVAL); the
 *   result is a database of one length; the source stays valid. */
int kmg_db_restrict(kmg_ctx *ctx, const kmg_db *db, int length, kmg_db **out);
/* kmg_db_restrict_range: the same stream function with min_length < max_length (Matcher -k a -K b on the search side,
 *   Matcher.java:224,241): k-mers shorter than min_length dropped, longer than max_length cut to it and folded, the
 *   others untouched; the result is a variable-length database declared [min_length, max_length] (the declared lengths
 *   are what the trie is built with, DB.java:86,154), whatever lengths it holds.  key_len <= min_length <=
 *   max_length <= 32. */
int kmg_db_restrict_range(kmg_ctx *ctx, const kmg_db *db, int min_length, int max_length, kmg_db **out);
/* kmg_match_trie: DB.getNearestKmers' non-quick branch with maxDiff = 0 (Database/DB.java:83,139-184) for a search stream
 *   of k-mers of several lengths -- what Matcher does on variable-length files when -k / -K are left at their defaults
 *   (Matcher.java:147-148): per key a radix trie over the database's k-mers (Root.addKmer, Root.java:29-113) and
 *   Root.closestKmers (:115-170) per search k-mer.  The trie's answer for a search k-mer t of T bases is the data object
 *   of the node covering depth T on t's path, which -- because a split node shares its object with its tail (:79-80) and
 *   k-mers merge into the nodes they pass from depth `shortest` on (:48-51,95-102) -- holds the summed counts of every
 *   stored k-mer sharing the first max(min_k, L + 1) bases of x0, the first stored k-mer with prefix t, L = the bases x0
 *   shares with its predecessor in the key.  Computed here in closed form, bit-exact against the literal restatement
 *   (tests).  min_k = the search stream's declared shortest length (Root's `shortest`); the stream's lengths
 *   must lie inside the database's (DB.java:78-81 -> KMG_EINVAL).  count_cap > 0: stored counts are cut to it before
 *   they are added (255 = what a database file holds, CountDataType.java:40; 0 for handles loaded from files).
 *   Out: hit_row[n_unique(search)] = row of the answer in *rows, or -1 (no stored k-mer has t as a prefix: no match
 *   line); *rows = a taxon database whose row r (kmg_db_fetch: val_off / val_a / val_b) is answer r, taxa ascending --
 *   its keys are row numbers, not k-mers.  The database searched is one file, a set laid end to end (kmg_db_merge of
 *   files that share no key), or a set merged from files that SHARE keys: the reference inserts the files of a set one
 *   after the other, unmerged (DB.java:154-158), so a later file's k-mer can split an edge an earlier file's k-mer made,
 *   both pieces then hold that k-mer's data object (Root.java:79-80) and a k-mer that walks through both is merged into
 *   it twice.  For such a database the answer is computed over the files kmg_db_merge kept, in their order, by the
 *   closed form for several files (match.h; tests/test_trie_closed_form.py holds the derivation to the literal trie).
 *   A SEARCH side merged from files that share keys returns KMG_EUNSUPPORTED: the reference does not merge its search
 *   stream (Matcher.java:241). */
int kmg_match_trie(kmg_ctx *ctx, const kmg_db *search, const kmg_db *db, int min_k, int count_cap, kmg_db **rows,
                   int64_t *hit_row, uint64_t *n_hits);
/* kmg_db_load : upload a decoded database (KmerFile.java:200-250 output) from host CSR arrays. */
int kmg_db_load(kmg_ctx *ctx, int k, int key_len, int rc, int payload_kind, const uint64_t *keys,
                uint64_t n_unique, const uint64_t *val_off, const int32_t *val_a, const int32_t *val_b,
                kmg_db **out);
/* Device-resident variants (no counterpart in the reference: they exist so that a multi-GPU caller can move
 * database slices between GPUs -- kmers_amd/dist.py, the key-range exchange of SURVEY 8e -- without staging
 * them through host memory).
 * kmg_db_copy_dev: copy elements [first, first + count) of one CSR array of `db` into caller-owned DEVICE memory
 *   (u64 for keys / val_off / bucket_off, int32 for val_a / val_b); synchronises before returning.
 * kmg_db_load_dev: kmg_db_load from DEVICE arrays (copied; the caller keeps ownership).  Checked on the device:
 *   keys strictly ascending, val_off running from 0 to n_values (KMG_EINVAL otherwise). */
#define KMG_DB_KEYS       0
#define KMG_DB_VAL_OFF    1
#define KMG_DB_VAL_A      2
#define KMG_DB_VAL_B      3
#define KMG_DB_BUCKET_OFF 4
#define KMG_DB_KEYS_LO    5   /* u64; databases of k > 32 only */
int kmg_db_copy_dev(const kmg_db *db, int which, void *d_dst, uint64_t first, uint64_t count);
int kmg_db_load_dev(kmg_ctx *ctx, int k, int key_len, int rc, int payload_kind, const uint64_t *d_keys,
                    uint64_t n_unique, const uint64_t *d_val_off, const int32_t *d_val_a, const int32_t *d_val_b,
                    uint64_t n_values, kmg_db **out);
void kmg_db_free(kmg_db *db);
/* kmg_db_encode: the per-key block bytes MakeAndWriteKey writes (FileCreator.java:312-347:
 *   [shared-prefix byte][rest of compressedBytes][CountDataType | SetCompressor<ReadPos> bytes]),
 *   all 4^key_len blocks back to back, computed on the GPU.  Two calls: first with out == NULL to
 *   get *n_bytes, then with a buffer of that size.  block_off[4^key_len + 1] (may be NULL). */
int kmg_db_encode(kmg_ctx *ctx, const kmg_db *db, uint8_t *out, uint64_t cap, uint64_t *n_bytes,
                  uint64_t *block_off);

/* kmg_db_merge: one database from several (replaces DB.kmers over a multi-file database, Database/DB.java:99-109
 *   + jar StreamUtils.mergeSortedStreams / groupAndReduceStream: k-way merge of the files' ascending k-mer
 *   streams, equal k-mers reduced with addAll -- per-taxon counts add (as long in Java; here int32, saturating),
 *   read-position sets unite).  Inputs must agree in datatype, k and key_len (DB.java:36-42,57-64 ->
 *   KMG_EINVAL); they stay valid.  Also the local step after a key-range exchange between GPUs.
 *   Variable-length databases (minK < maxK): a set whose files share no key -- the sets MakeDatabase -S / -G write are
 *   cut at key bounds -- is laid end to end in key order; files that share keys are merged in Sequence.compareTo order
 *   (zero-padded value, then length: Kmers/Sequence.java:165-187), length by length.  Such a merged database serves
 *   TaxaCounts, kmg_db_restrict and the exact join behind it, and -- because the reference's trie takes a key's k-mers
 *   file by file, unmerged (DB.java:154-158), so that its answers depend on the files and their order -- it keeps copies
 *   of its (taxon) files, in the order they were given, for kmg_match_trie. */
int kmg_db_merge(kmg_ctx *ctx, const kmg_db *const *dbs, int n_dbs, kmg_db **out);

/* ---- stage 3: match  (replaces DB.getNearestKmers quick branch, DB.java:76-90,129-137) -------
 * Sorted inner join of the search database's distinct k-mers against the reference database:
 * hit_index[i] = index (into the reference CSR) of the k-mer equal to search k-mer i, else -1.
 * Same k and key_len required (DB.java:78-81 -> KMG_EINVAL).  n_hits may be NULL.
 * kmg_match_dev writes hit_index to a device pointer and does not synchronise.  The databases may belong to
 * another context of the same GPU (a caller may build the reference and the reads side by side on two contexts);
 * lifetime rule: they must stay alive until the call returns -- after that kmg_db_free is safe at once, the library
 * orders the release of a foreign database's memory behind the join that still reads it. */
int kmg_match(kmg_ctx *ctx, const kmg_db *search, const kmg_db *ref, int64_t *hit_index, uint64_t *n_hits);
int kmg_match_dev(kmg_ctx *ctx, const kmg_db *search, const kmg_db *ref, int64_t *d_hit_index);

/* kmg_match_keys_dev: the same join for search k-mers that are not a database of this library: d_keys (device)
 *   holds n_runs (1 .. 64) ascending runs back to back, run i = elements [run_off[i], run_off[i+1]) (run_off: HOST
 *   array of n_runs + 1, non-decreasing).  That is what the owner of a key range holds after the key-range exchange
 *   of a multi-GPU search (the reference opens such a sharded reference as the file set name.1 .. name.N,
 *   Matcher.java:80-89, and walks its k-mers in ascending order, DB.java:99-109): one ascending slice of read
 *   k-mers from every GPU, all inside the owner's range -- joined in ONE launch, hit_index[i] for element i, written
 *   to a device pointer without synchronising.  k = length of the search k-mers (DB.java:78-81 -> KMG_EINVAL). */
int kmg_match_keys_dev(kmg_ctx *ctx, const uint64_t *d_keys, const uint64_t *run_off, int n_runs, int k,
                       const kmg_db *ref, int64_t *d_hit_index);
/* kmg_match_keys_wide_dev: the same for two-word k-mers (33 <= k <= 64; Sequence.java:15-35,90-115): d_keys holds the first
 *   words, d_keys_lo the second, run for run; every run is joined by a launch of its own. */
int kmg_match_keys_wide_dev(kmg_ctx *ctx, const uint64_t *d_keys, const uint64_t *d_keys_lo, const uint64_t *run_off,
                            int n_runs, int k, const kmg_db *ref, int64_t *d_hit_index);

/* kmg_match_dev_noalloc: kmg_match_dev for callers that cannot allocate.  No counterpart in the reference: it is
 *   for code under HIP graph capture, tight serving loops and harnesses that forbid hipMalloc.  Same result as
 *   kmg_match_dev, to the bit; calls no hipMalloc / hipFree / allocating routine and does not synchronise.  Its only
 *   scratch, the per-tile bounds of the join, lives in a file-scope device array of 1 MiB that every GPU the library
 *   runs on holds until the process ends; hence the limit: at most KMG_MATCH_NOALLOC_MAX_SEARCH distinct search
 *   k-mers per call (KMG_EINVAL beyond).  Both databases must belong to ctx.  Every call on a GPU shares that GPU's
 *   array; calls from several contexts of one GPU (= several streams) are ordered by the library, each behind the one
 *   queued before it (an event per context, made with the context), so they give the right results but do not
 *   overlap.  Two-word k-mers: their join finds each tile's piece of the reference inside the kernel, so the array is
 *   not touched (and nothing is waited for); the limit is the same. */
#define KMG_MATCH_NOALLOC_MAX_SEARCH 268433408ull   /* 131071 tiles x 2048 k-mers */
int kmg_match_dev_noalloc(kmg_ctx *ctx, const kmg_db *search, const kmg_db *ref, int64_t *d_hit_index);

/* kmg_match_near: inexact search for k-mers of one length (replaces DB.getNearestKmers' other branch ->
 *   processNearestCommonKey, DB.java:139-184, over Root.addKmers / closestKmers, Root.java:22-170, and
 *   KmerUtils.getCloseKeys, KmerUtils.java:18-47; Matcher -n 1 | 2): for every search k-mer ALL reference k-mers
 *   within Hamming distance max_diff (any; up to 2 the neighbours are enumerated, beyond that the trie walk of
 *   Root.closestKmers, Root.java:115-170, runs over the sorted array: cost grows as in the reference; two-word k-mers take the walk for every
 *   max_diff), as pairs (ref_index, dist) in ascending
 *   reference order -- the order the reference's trie hands them to ClosestInfoCollector.add -- with
 *   pair_off[i] .. pair_off[i + 1] the pairs of search k-mer i.  `-j` (only the best distance), the reverse-complement
 *   rule of ClosestInfoCollector.add (:30-50) and the KmerDiff text (KmerDiff.java:54-67) are cheap per-k-mer steps
 *   left to the caller.  Two calls: ref_index == dist == NULL -> *n_pairs (and pair_off[n_unique + 1] if given); then
 *   with arrays of `capacity` >= *n_pairs (the sizing call leaves its offsets with the search database, which must
 *   belong to ctx, so the second call only fills). */
int kmg_match_near(kmg_ctx *ctx, const kmg_db *search, const kmg_db *ref, int max_diff, uint64_t *pair_off,
                   int64_t *ref_index, uint8_t *dist, uint64_t capacity, uint64_t *n_pairs);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI (no counterpart in the reference, which is one process) ------
 * The reference's form of a database in pieces is the file set name.1 .. name.N (Matcher.java:80-89), walked in ascending
 * k-mer order (DB.kmers, DB.java:99-109).  Here piece g is the contiguous range of key buckets [bounds[g], bounds[g+1])
 * (Kmer.key(key_len), Kmer.java:47-55) that GPU g owns; bounds has nranks + 1 ascending entries in 0 .. 4^key_len.
 * kmg_comm_unique_id : rank 0 makes the 128-byte id (ncclGetUniqueId) and hands it to the other ranks by any means.
 * kmg_comm_create    : every rank, with the context whose GPU and stream the exchange uses (ncclCommInitRank).  RCCL is
 *                      looked up at run time; KMG_EUNSUPPORTED where librccl.so is absent.
 * kmg_match_sharded  : the sharded Matcher join.  Collective: every rank calls it with its own search database and its
 *                      own shard of the reference.  The search k-mers are cut at the bounds (a database is sorted: each
 *                      owner's part is a slice), exchanged by grouped ncclSend / ncclRecv, joined at the owners in one
 *                      launch (kmg_match_keys_dev's kernels) and the hit indices exchanged back:
 *                      d_hit_index[i] (DEVICE, n_unique of `search`) = index INTO THE OWNER'S SHARD of the k-mer equal to
 *                      search k-mer i, -1 = none; the owner of k-mer i is the range its bucket falls in.  Asynchronous on
 *                      kmg_stream() after two small host round trips (cut points, counts).
 * kmg_db_exchange_by_key_range: for inputs that are sharded too -- every rank holds the database of ITS part of the
 *                      input; each range's slice travels to its owner (four all-to-all-v: k-mers, values per k-mer, two
 *                      value arrays) and the owner merges what arrived as kmg_db_merge does.  *out = this rank's shard.
 *                      Two-word k-mers: both calls move the second words beside the first (one more all-to-all-v). */
typedef struct kmg_comm kmg_comm;
#define KMG_COMM_ID_BYTES 128
int  kmg_comm_unique_id(uint8_t *id /* KMG_COMM_ID_BYTES */);
int  kmg_comm_create(kmg_ctx *ctx, int nranks, int rank, const uint8_t *id, kmg_comm **out);
void kmg_comm_destroy(kmg_comm *comm);
int  kmg_comm_info(const kmg_comm *comm, int *nranks, int *rank);
int  kmg_match_sharded(kmg_comm *comm, const kmg_db *search, const kmg_db *ref_shard, const uint32_t *bounds,
                       int64_t *d_hit_index);
int  kmg_db_exchange_by_key_range(kmg_comm *comm, const kmg_db *local, const uint32_t *bounds, kmg_db **out);

/* ---- TaxaCounts default mode (replaces TaxaCounts.java:119-130,355-360,386-395) --------------
 * Over every stored k-mer: match[t]++ for each taxon of its list, unique[t]++ if the list has
 * exactly one taxon.  Output ascending by taxon.  In: *n_taxa = capacity; out: taxa found
 * (if larger than the capacity only the first capacity rows are written). */
int kmg_taxacounts(kmg_ctx *ctx, const kmg_db *db, int32_t *taxon, int64_t *match, int64_t *unique,
                   uint64_t *n_taxa);

/* ---- per-read aggregation downstream of the exact match ---------------------
 * Both join `search` (a read database) against `ref` (a reference database) themselves.
 * kmg_collect_by_read: Utils/CollectByRead.java:159-184 (processKmer) + :105-124 (ProcessIndex): every
 *   (read, pos) occurrence of every search k-mer that has a match, in ReadPos order (Reads/ReadPos.java:27-35:
 *   read, then pos as a signed short), each with the index of the reference k-mer it hit -- the line
 *   `read:pos \t 0:TxC|TxC` of the reference's output is (read[i], pos[i]) + the taxa of ref k-mer ref_index[i]
 *   (counts capped at 255 as CountDataType.compress does on the way through its temp file).  The reference keys
 *   its output blocks by (read / 1000) * 1000; the caller cuts the arrays there.  read == NULL: only *n_out.
 *   Capacity of the three arrays: n_values of `search` always suffices.  Where two sequences share a read index
 *   one (read, pos) can occur twice; both records are returned, smaller k-mer first (the reference's TreeSet
 *   keeps whichever it meets first, in an order it does not define).
 * kmg_classify_reads: Utils/ReadClassifier.java:178-220 (processRead) over the same records, without
 *   materialising them: call[r] = the one taxon t >= 0 such that every single-taxon hit of read r names t and
 *   every hit's taxon set contains t, else -1 (no line in the reference's output).  n_reads = number of read
 *   indices (KMG_EINVAL if the database holds a larger one).
 * kmg_classify_reads_hits: the same, and has_hit[r] = 1 if read r has at least one matched k-mer -- what decides
 *   whether the reference's output holds a block for (r / 1000) * 1000 at all (ReadClassifier.java:157-160 writes
 *   one block per block of CollectByRead's output, and CollectByRead.java:105-124 one per group of reads with a
 *   record), so a caller can write ReadClassifier's file straight from two databases.
 * kmg_classify_reads_rows: the same calls from answers found by another search -- `rows` and hit_row as
 *   kmg_match_trie hands them out for `search` (the non-quick branch of DB.getNearestKmers, DB.java:139-184: what
 *   Matcher does with its defaults on default databases), hit_row a host array of n_kmers(search) entries: the
 *   taxa of search k-mer i are those of row hit_row[i], -1 = no match line.  KMG_EINVAL for an index outside the rows. */
int kmg_collect_by_read(kmg_ctx *ctx, const kmg_db *search, const kmg_db *ref, uint64_t *n_out, int32_t *read,
                        int32_t *pos, int64_t *ref_index);
int kmg_classify_reads(kmg_ctx *ctx, const kmg_db *search, const kmg_db *ref, uint64_t n_reads, int32_t *call);
int kmg_classify_reads_hits(kmg_ctx *ctx, const kmg_db *search, const kmg_db *ref, uint64_t n_reads, int32_t *call,
                            uint8_t *has_hit);
int kmg_classify_reads_rows(kmg_ctx *ctx, const kmg_db *search, const kmg_db *rows, const int64_t *hit_row,
                            uint64_t n_reads, int32_t *call, uint8_t *has_hit);

/* ---- measurement hook (bench.py) --------------------------------------------------------------
 * Total device time in ms, measured with HIP events on kmg_stream(), of the launches of one
 * kernel (ids 0 .. kmg_prof_count()-1, names from kmg_prof_name) since the last kmg_prof_reset,
 * their launch count and the ALGORITHMIC bytes those launches had to move.
 * Profiling is off by default (events cost a little): enable with kmg_prof_enable(ctx, 1). */
int kmg_prof_enable(kmg_ctx *ctx, int on);
int kmg_prof_reset(kmg_ctx *ctx);
int kmg_prof_get(kmg_ctx *ctx, int kernel_id, double *total_ms, uint64_t *launches, uint64_t *algo_bytes);
/* Launches calib_copy_kernel: an 8-byte-per-lane streaming copy of n_words words whose HBM traffic is
 * known exactly (8n read + 8n written) -- the calibration launch for the rocprofv3 FETCH_SIZE /
 * WRITE_SIZE counters (tools/pmc_traffic.py). */
int kmg_calib_copy(kmg_ctx *ctx, uint64_t n_words);
int kmg_prof_count(void);               /* number of kernel ids */
const char *kmg_prof_name(int kernel_id);

#ifdef __cplusplus
}
#endif
#endif /* KMERS_GPU_H */
