// host_parse.cpp -- host-side record splitting of FASTA/FASTQ text into sequence batches.
//
// This is the part of OtherFiles/KmersFromFile.java that stays on the host (north star: parsing stays
// with the caller): the ID / KMER / OTHER state machine of KmersFromFileStateChanger.java:46-68 run
// over the file's characters.  It produces what kmg_build_add consumes: for every entry into the KMER
// state one "sequence" = the characters > 32 seen until the state is left (KmersFromFile.java:154-158:
// only those advance `pos`), plus the header id current at that time.  Windows, validity, RC and read
// numbering are the GPU's job.  Plain C++ (no HIP); built into libkmers_host.so.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

namespace {
enum { S_ID = 0, S_KMER = 1, S_OTHER = 2 };

struct Batch {
    std::unique_ptr<uint8_t[]> bases;   // not value-initialised: only what is written gets touched
    uint64_t nbases = 0;
    std::vector<uint64_t> seq_off;   // nseq + 1
    std::vector<int32_t> seq_id;     // FASTA: Integer.parseInt(header id); 0 where unparsable
    std::vector<uint8_t> id_ok;      // FASTA: 1 if the header parsed as a Java int
    std::vector<int64_t> seq_tok;    // FASTQ: token of the header string (equal <=> equal strings)
    std::vector<uint64_t> name_off;  // token t's header = names[name_off[t] .. name_off[t+1])
    std::string names;
    bool per_seq_names = false;      // a stream's piece: names / name_off go by sequence, not by token
    int eof_in_kmer = 0;             // file ended inside a sequence line whose last char was a base
    int high_bytes = 0;              // a sequence character >= 128 was seen (see kmh_high_bytes)
    int empty_header = 0;            // FASTQ: a record with an empty header line was seen (see kmh_empty_header)
    std::string err;
};

// Header strings -> tokens (equal <=> equal strings).  A flat open-addressing table over the names already stored in
// the batch: one probe and one short memcmp per header (a node-based map spent three quarters of the parse here).
struct NameTable {
    struct Slot { uint64_t hash; int64_t tok; };       // tok < 0: empty
    std::vector<Slot> slots;
    size_t used = 0;
    NameTable() : slots(1 << 16, Slot{0, -1}) {}
    static uint64_t hash_of(const uint8_t *p, size_t n) {
        uint64_t h = 0x9E3779B97F4A7C15ull ^ (n * 0xD6E8FEB86659FD93ull);
        while (n >= 8) {
            uint64_t w;
            memcpy(&w, p, 8);
            h = (h ^ w) * 0xD6E8FEB86659FD93ull;
            h ^= h >> 32;
            p += 8; n -= 8;
        }
        uint64_t w = 0;
        memcpy(&w, p, n);
        h = (h ^ w) * 0xD6E8FEB86659FD93ull;
        return h ^ (h >> 29);
    }
    void grow() {
        std::vector<Slot> old(slots.size() * 2, Slot{0, -1});
        old.swap(slots);
        const size_t mask = slots.size() - 1;
        for (const Slot &s : old)
            if (s.tok >= 0) {
                size_t i = s.hash & mask;
                while (slots[i].tok >= 0) i = (i + 1) & mask;
                slots[i] = s;
            }
    }
    // token of the string, appended to names / name_off if it is new
    int64_t intern(const uint8_t *p, size_t n, std::string &names, std::vector<uint64_t> &name_off) {
        if (used * 2 >= slots.size()) grow();
        const uint64_t h = hash_of(p, n);
        const size_t mask = slots.size() - 1;
        size_t i = h & mask;
        for (; slots[i].tok >= 0; i = (i + 1) & mask) {
            if (slots[i].hash != h) continue;
            const uint64_t a = name_off[slots[i].tok], e = name_off[slots[i].tok + 1];
            if (e - a == n && memcmp(names.data() + a, p, n) == 0) return slots[i].tok;
        }
        const int64_t tok = (int64_t)name_off.size() - 1;
        names.append((const char *)p, n);
        name_off.push_back(names.size());
        slots[i] = Slot{h, tok};
        used++;
        return tok;
    }
};

// Integer.parseInt: optional sign, at least one digit, fits in int32.
bool parse_java_int(const uint8_t *p, size_t n, int32_t *out) {
    if (n == 0) return false;
    size_t i = 0;
    bool neg = false;
    if (p[0] == '-' || p[0] == '+') { neg = p[0] == '-'; i = 1; }
    if (i == n) return false;
    int64_t v = 0;
    for (; i < n; i++) {
        if (p[i] < '0' || p[i] > '9') return false;
        v = v * 10 + (p[i] - '0');
        if (v > 2147483648LL) return false;
    }
    if (neg) v = -v;
    if (v > 2147483647LL || v < -2147483648LL) return false;
    *out = (int32_t)v;
    return true;
}
}  // namespace

extern "C" {

typedef struct kmh_batch kmh_batch;

}  // extern "C"

namespace {
// What a stream keeps between its pieces (kmh_stream_*): header strings -> tokens over the WHOLE input, so that equal
// tokens mean equal strings across pieces (read numbering compares a read with the last one that handed out a k-mer).
struct Stream {
    int mode = 1;
    NameTable intern;
    std::string names;
    std::vector<uint64_t> name_off{0};
    std::string carry;           // input from the last header start on that the previous piece left unparsed
};

// The splitter proper.  Whole input: `st` = NULL, tokens index b->names / b->name_off.  One piece of a stream: tokens come
// from st's table, b->names / b->name_off hold the header text PER SEQUENCE (name_off has nseq + 1 entries), and unless
// `final` everything from the last header start on is rolled back and left to the next piece: *consumed = its offset.
Batch *parse_core(const uint8_t *data, uint64_t n, int mode, int k, Stream *st, bool final, uint64_t *consumed) {
    Batch *b = new Batch();
    int8_t trans[3][256];
    memset(trans, -1, sizeof trans);
    const bool named = mode == 2;
    if (named) mode = 0;
    if (mode == 0) {
        trans[S_ID][' '] = S_OTHER; trans[S_ID]['\n'] = S_KMER; trans[S_OTHER]['\n'] = S_KMER;
        trans[S_KMER]['\n'] = S_OTHER; trans[S_OTHER]['>'] = S_ID;
    } else {
        trans[S_OTHER]['@'] = S_ID; trans[S_KMER]['@'] = S_ID; trans[S_ID]['\n'] = S_KMER;
        trans[S_KMER]['+'] = S_OTHER;
    }
    NameTable own_table;
    b->per_seq_names = st != nullptr;
    NameTable &intern = st ? st->intern : own_table;
    std::string &tnames = st ? st->names : b->names;
    std::vector<uint64_t> &tname_off = st ? st->name_off : b->name_off;
    b->name_off.push_back(0);
    // the last header start and what the batch held then (stream pieces are cut there)
    uint64_t cut_i = 0, cut_nout = 0;
    size_t cut_nseq = 0, cut_names = 0;
    bool have_cut = false;
    int state = S_OTHER;
    uint64_t id_start = 0, id_len = 0;
    int32_t cur_id = 0; uint8_t cur_ok = 0; int64_t cur_tok = -1;
    bool id_done = false, in_seq = false, last_was_base = false;
    auto finish_id = [&]() {
        if (id_done) return;
        id_done = true;
        if (mode == 0 && !named) {
            cur_ok = parse_java_int(data + id_start, id_len, &cur_id) ? 1 : 0;
            if (!cur_ok) cur_id = 0;
        } else if (named) {
            cur_tok = intern.intern(data + id_start, id_len, tnames, tname_off);
        } else {
            cur_tok = intern.intern(data + id_start, id_len, tnames, tname_off);
            if (id_len == 0) b->empty_header = 1;
        }
    };
    // stream pieces: the header text of the sequence being opened, one entry per sequence
    auto note_name = [&]() {
        if (!st || (mode == 0 && !named)) return;
        b->names.append((const char *)data + id_start, id_len);
        b->name_off.push_back(b->names.size());
    };
    b->seq_off.push_back(0);   // seq_off = [0, end_0, end_1, ...]: a sequence is opened/closed in turn
    // bases are written through a raw pointer into a buffer sized for the whole input (it can only shrink)
    b->bases.reset(new uint8_t[n + (k > 0 ? (uint64_t)k : 0) + 1]);   // + room for the repeated last window (below)
    uint8_t *out = b->bases.get();
    uint64_t nout = 0;
    auto close_seq_fast = [&]() {
        if (in_seq) { b->seq_off.push_back(nout); in_seq = false; }
    };
    // is there any transition out of `st` on character c?
    bool special[3][256];
    for (int st = 0; st < 3; st++)
        for (int c = 0; c < 256; c++) special[st][c] = trans[st][c] >= 0;
    // Whole lines at a time where nothing special happens (the bulk of any FASTA / FASTQ file): the characters of
    // a state that have a transition, for the states that have one or two of them
    int nsp[3] = {0, 0, 0};
    uint8_t spc[3][2] = {{0, 0}, {0, 0}, {0, 0}};
    for (int st = 0; st < 3; st++)
        for (int c = 0; c < 256; c++)
            if (special[st][c]) { if (nsp[st] < 2) spc[st][nsp[st]] = (uint8_t)c; nsp[st]++; }
    uint64_t i = 0;
    while (i < n) {
        const uint8_t c = data[i];
        const int ns = trans[state][c];
        if (ns >= 0) {
            if (state == S_KMER) close_seq_fast();
            state = ns;
            last_was_base = false;
            if (state == S_ID) {
                id_start = i + 1; id_len = 0; id_done = false;
                if (st) {      // (a sequence is closed by now: leaving S_KMER closed it)
                    have_cut = true; cut_i = i; cut_nout = nout; cut_nseq = b->seq_off.size(); cut_names = b->names.size();
                }
            }
            else if (state == S_KMER) {
                finish_id();
                note_name();
                // open a new sequence (pos = 0, KmersFromFile.java:137-139)
                in_seq = true;
                if (mode == 0 && !named) { b->seq_id.push_back(cur_id); b->id_ok.push_back(cur_ok); }
                else b->seq_tok.push_back(cur_tok);
            }
            i++;
            continue;
        }
        // a run of characters without a transition: handled in one go per state
        const bool *sp = special[state];
        uint64_t j = i + 1;
        if (state == S_KMER && c > 32 && nsp[S_KMER] <= 2) {
            // up to the end of the line: if it holds nothing but bases, copy it in one piece
            const uint8_t *nl = (const uint8_t *)memchr(data + i, '\n', n - i);
            const uint64_t end = nl ? (uint64_t)(nl - data) : n;
            const uint8_t s1 = spc[S_KMER][0], s2 = nsp[S_KMER] == 2 ? spc[S_KMER][1] : spc[S_KMER][0];
            unsigned odd = 0, high = 0;
            for (uint64_t q = i; q < end; q++) {
                odd |= (unsigned)(data[q] <= 32) | (unsigned)(data[q] == s1) | (unsigned)(data[q] == s2);
                high |= data[q];
            }
            if (!odd) {
                b->high_bytes |= (int)(high >> 7);
                memcpy(out + nout, data + i, end - i);
                nout += end - i;
                last_was_base = true;
                i = end;
                continue;
            }
        } else if (state != S_KMER && nsp[state] == 1) {
            // header or quality text: straight to the one character that ends it
            const uint8_t *at = (const uint8_t *)memchr(data + i, spc[state][0], n - i);
            j = at ? (uint64_t)(at - data) : n;
            if (state == S_ID) id_len += j - i;
            i = j;
            continue;
        }
        if (state == S_KMER) {
            // sequence characters: everything > 32 is a base position (KmersFromFile.java:154-158)
            if (c > 32) { out[nout++] = c; last_was_base = true; b->high_bytes |= c >> 7; } else last_was_base = false;
            while (j < n && !sp[data[j]]) {
                const uint8_t d = data[j++];
                if (d > 32) { out[nout++] = d; last_was_base = true; b->high_bytes |= d >> 7; } else last_was_base = false;
            }
        } else {
            while (j < n && !sp[data[j]]) j++;
            if (state == S_ID) id_len += j - i;
        }
        i = j;
    }
    if (st && !final) {
        // leave the record that was being read to the next piece
        if (!have_cut) { *consumed = 0; b->seq_off.assign(1, 0); b->seq_id.clear(); b->id_ok.clear(); b->seq_tok.clear();
                         b->names.clear(); b->name_off.assign(1, 0); b->nbases = 0; return b; }
        b->seq_off.resize(cut_nseq);
        const size_t ns = cut_nseq - 1;
        if (mode == 0 && !named) { b->seq_id.resize(ns); b->id_ok.resize(ns); }
        else { b->seq_tok.resize(ns); b->names.resize(cut_names); b->name_off.resize(ns + 1); }
        b->nbases = cut_nout;
        *consumed = cut_i;
        return b;
    }
    if (consumed) *consumed = n;
    b->nbases = nout;
    auto close_seq = [&]() {
        if (in_seq) { b->seq_off.push_back(b->nbases); in_seq = false; }
    };
    const bool ended_in_kmer = state == S_KMER && in_seq && last_was_base;
    close_seq();
    const size_t nseq = b->seq_off.size() - 1;
    if (ended_in_kmer) {
        b->eof_in_kmer = 1;
        const uint64_t last_len = b->seq_off[nseq] - b->seq_off[nseq - 1];
        if (mode == 0 && k > 0 && last_len >= (uint64_t)k) {
            const uint64_t from = b->nbases - (uint64_t)k;
            for (int j = 0; j < k; j++) out[b->nbases + j] = out[from + j];
            b->nbases += (uint64_t)k;
            b->seq_off.push_back(b->nbases);
            if (named) {
                b->seq_tok.push_back(b->seq_tok.back());
                if (st) {
                    const size_t q = b->name_off.size();
                    b->names.append(b->names, b->name_off[q - 2], b->name_off[q - 1] - b->name_off[q - 2]);
                    b->name_off.push_back(b->names.size());
                }
            } else {
                b->seq_id.push_back(b->seq_id.back());
                b->id_ok.push_back(b->id_ok.back());
            }
        }
    }
    return b;
}
}  // namespace

extern "C" {

// mode 0 = FASTA (KmersFromFileStateChanger.getFAinstance), 1 = FASTQ (getFQinstance), 2 = FASTA whose header ids
// are kept as strings (tokens + names, as for FASTQ) for kmh_apply_id_map (MakeDatabase -m).
// k is only used for the no-trailing-newline quirk, FASTA only: when the file ends
// inside a sequence line right after a base, the reference emits the last k-mer twice; this appends
// the last k characters as one more sequence with the same id, which yields exactly that extra k-mer
// when (and only when) the last window is valid.
kmh_batch *kmh_parse(const uint8_t *data, uint64_t n, int mode, int k) {
    return (kmh_batch *)parse_core(data, n, mode, k, nullptr, true, nullptr);
}

// The same input in pieces of any size (inputs that should not sit in memory whole, and readers that overlap the file
// with the GPU): kmh_stream_next parses what the previous piece left plus `data`, up to the last header start it sees
// (all of it when `final`), and returns the sequences of that part as a batch of their own.  Tokens are equal across
// pieces iff the header strings are; kmh_name_off / kmh_names of such a batch hold the header text PER SEQUENCE
// (kmh_nseq + 1 offsets), not per token.  The batches of a stream, one after the other, are the sequences kmh_parse
// finds in the whole input (tests/test_host_parse.py cuts every case at every size).
typedef struct kmh_stream kmh_stream;
kmh_stream *kmh_stream_open(int mode) {
    Stream *st = new Stream();
    st->mode = mode;
    return (kmh_stream *)st;
}
void kmh_stream_close(kmh_stream *h) { delete (Stream *)h; }
kmh_batch *kmh_stream_next(kmh_stream *h, const uint8_t *data, uint64_t n, int final, int k) {
    Stream *st = (Stream *)h;
    uint64_t used = 0;
    Batch *b;
    if (st->carry.empty()) {
        b = parse_core(data, n, st->mode, k, st, final != 0, &used);
        if (!final) st->carry.assign((const char *)data + used, n - used);
    } else {
        if (!final && !memchr(data, st->mode == 1 ? '@' : '>', n)) {
            // No header can start in this piece, so nothing new is complete: the open record only grows.  (Parsing the
            // carry again from its start for every such piece cost O(record^2 / piece) on a chromosome-sized record --
            // ADVICE r2.  The open record itself still waits in memory whole: its sequences leave with the record.)
            st->carry.append((const char *)data, n);
            b = new Batch();
            b->seq_off.assign(1, 0);
            b->name_off.assign(1, 0);
            b->per_seq_names = true;
            return (kmh_batch *)b;
        }
        st->carry.append((const char *)data, n);
        b = parse_core((const uint8_t *)st->carry.data(), st->carry.size(), st->mode, k, st, final != 0, &used);
        if (!final) st->carry.erase(0, used); else st->carry.clear();
    }
    return (kmh_batch *)b;
}
// bytes the stream holds back for its next piece (a reader sizes its next read by it)
uint64_t kmh_stream_pending(const kmh_stream *h) { return ((const Stream *)h)->carry.size(); }

// MakeDatabase -m (MakeDatabase.java:138-147): `map` = the text of the map file, one "<sequence id>\t<taxon id>" per
// line (BufferedReader.lines: \n, \r\n or \r end a line; String.split("\t") -> parts[0], Integer.parseInt(parts[1]);
// a later line replaces an earlier one with the same id, HashMap.put).  For a batch parsed in mode 2 this fills seq_id /
// id_ok as mode 0 does: id_ok = 0 where the header is not in the map (KmersFromFile.java:46 hands out null there).
// Returns 0, or the 1-based number of the first line the reference would throw on (no tab, or not an int).
uint64_t kmh_apply_id_map(kmh_batch *h, const uint8_t *map, uint64_t n) {
    Batch *b = (Batch *)h;
    std::unordered_map<std::string, int32_t> ids;
    uint64_t line = 0, i = 0;
    while (i < n) {
        uint64_t e = i;
        while (e < n && map[e] != '\n' && map[e] != '\r') e++;
        line++;
        const uint8_t *t1 = (const uint8_t *)memchr(map + i, '\t', e - i);
        if (!t1) return line;                              // parts[1]: ArrayIndexOutOfBoundsException
        const uint8_t *t2 = (const uint8_t *)memchr(t1 + 1, '\t', (map + e) - (t1 + 1));
        const uint8_t *vend = t2 ? t2 : map + e;
        int32_t v = 0;
        if (!parse_java_int(t1 + 1, (size_t)(vend - (t1 + 1)), &v)) return line;   // (an empty parts[1] included)
        ids[std::string((const char *)map + i, (size_t)(t1 - (map + i)))] = v;
        i = e;
        if (i < n) i += (map[i] == '\r' && i + 1 < n && map[i + 1] == '\n') ? 2 : 1;
    }
    const size_t ntok = b->name_off.size() - 1, nseq = b->seq_tok.size();   // (a stream's piece: one name per sequence)
    std::vector<int32_t> tok_id(ntok, 0);
    std::vector<uint8_t> tok_ok(ntok, 0);
    for (size_t t = 0; t < ntok; t++) {
        auto it = ids.find(b->names.substr(b->name_off[t], b->name_off[t + 1] - b->name_off[t]));
        if (it != ids.end()) { tok_id[t] = it->second; tok_ok[t] = 1; }
    }
    b->seq_id.resize(nseq);
    b->id_ok.resize(nseq);
    for (size_t s2 = 0; s2 < nseq; s2++) {
        const size_t t = b->per_seq_names ? s2 : (size_t)b->seq_tok[s2];
        b->seq_id[s2] = tok_id[t];
        b->id_ok[s2] = tok_ok[t];
    }
    return 0;
}

void kmh_free(kmh_batch *h) { delete (Batch *)h; }
uint64_t kmh_nseq(const kmh_batch *h) { return ((const Batch *)h)->seq_off.size() - 1; }
uint64_t kmh_nbases(const kmh_batch *h) { return ((const Batch *)h)->nbases; }
const uint8_t *kmh_bases(const kmh_batch *h) { return ((const Batch *)h)->bases.get(); }
const uint64_t *kmh_seq_off(const kmh_batch *h) { return ((const Batch *)h)->seq_off.data(); }
const int32_t *kmh_seq_id(const kmh_batch *h) { return ((const Batch *)h)->seq_id.data(); }
const uint8_t *kmh_id_ok(const kmh_batch *h) { return ((const Batch *)h)->id_ok.data(); }
const int64_t *kmh_seq_tok(const kmh_batch *h) { return ((const Batch *)h)->seq_tok.data(); }
uint64_t kmh_ntok(const kmh_batch *h) { return ((const Batch *)h)->name_off.size() - 1; }
const uint64_t *kmh_name_off(const kmh_batch *h) { return ((const Batch *)h)->name_off.data(); }
const char *kmh_names(const kmh_batch *h) { return ((const Batch *)h)->names.data(); }
int kmh_eof_in_kmer(const kmh_batch *h) { return ((const Batch *)h)->eof_in_kmer; }
// 1 if a sequence line held a byte >= 128.  The reference reads such a window by the sign of its FIRST byte
// (Sequence.java:15-33: `chars[0] > 4` on signed bytes -- a window that starts with a negative byte takes the "already
// codes" branch and is a k-mer of garbage codes iff all of its bytes are negative); the GPU path treats every such byte
// as an invalid base, so the drivers refuse input that holds one instead of building something different.
int kmh_high_bytes(const kmh_batch *h) { return ((const Batch *)h)->high_bytes; }
// 1 if a FASTQ record had an empty header.  ReadIDMapping starts with curReadS = "" (ReadIDMapping.java:10-16), so a
// first emitting read whose header is empty gets index -1 and no name line there; the GPU numbering starts at 0.  The
// drivers refuse such input instead of numbering it differently.
int kmh_empty_header(const kmh_batch *h) { return ((const Batch *)h)->empty_header; }


// ---- KmerFile decode (KmerFiles/KmerFile.java:200-250, CompressedKmerSpliterator.tryAdvance) ----------
// `data` = the bytes of every key block in ascending key order, back to back (each block's first record
// has copy == 0, so the prefix state needs no reset between blocks).  kind 0: CountDataType values
// ([short n][n x (int taxon, byte count)], Counts/CountDataType.java:83-95); kind 1: SetCompressor<ReadPos>
// ([int n][n x (int read, short pos)], jar + Reads/ReadPosDataType.java:41-44).  k-mers longer than 32
// bases cannot be represented on the GPU path: returns NULL.
struct Decoded {
    std::vector<uint64_t> keys, val_off;
    std::vector<uint64_t> keys_lo;      // bases 32 .. of k-mers of more than 32 bases (zero for the others)
    std::vector<int32_t> val_a, val_b;
    std::vector<uint8_t> lens;          // length byte of every k-mer (Kmer.compressedBytes, Kmer.java:90-97)
};

kmh_batch *kmh_decode_blocks(const uint8_t *data, uint64_t n, int kind) {
    Decoded *d = new Decoded();
    d->val_off.push_back(0);
    uint8_t prev[18] = {0}, kb[18];
    uint64_t off = 0;
    auto fail = [&]() -> kmh_batch * { delete d; return nullptr; };
    while (off < n) {
        int copy = (int8_t)data[off++];
        int len;
        if (copy == 0) { if (off >= n) return fail(); len = (int8_t)data[off++]; }
        else len = (int8_t)prev[0];
        if (len < 1 || len > 64 || copy < 0) return fail();
        const int l = (len - 1) / 4 + 1;
        if (copy > l + 1) return fail();
        if (copy > 0) {
            memcpy(kb, prev, (size_t)copy);
            const int need = l - copy + 1;
            if (off + need > n) return fail();
            memcpy(kb + copy, data + off, (size_t)need);
            off += need;
        } else {
            kb[0] = (uint8_t)len;
            if (off + l > n) return fail();
            memcpy(kb + 1, data + off, (size_t)l);
            off += l;
        }
        memcpy(prev, kb, (size_t)l + 1);
        uint64_t key = 0, key_lo = 0;
        for (int i = 0; i < l && i < 8; i++) key |= (uint64_t)kb[1 + i] << (56 - 8 * i);
        for (int i = 8; i < l; i++) key_lo |= (uint64_t)kb[1 + i] << (56 - 8 * (i - 8));
        d->keys.push_back(key);
        d->keys_lo.push_back(key_lo);
        d->lens.push_back((uint8_t)len);
        if (kind == 0) {
            if (off + 2 > n) return fail();
            const int cnt = (int16_t)((data[off] << 8) | data[off + 1]);
            off += 2;
            if (cnt < 0 || off + 5ull * cnt > n) return fail();
            for (int i = 0; i < cnt; i++) {
                const uint8_t *p = data + off;
                d->val_a.push_back((int32_t)(((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]));
                d->val_b.push_back((int32_t)p[4]);       // stored byte, 0..255 (CountDataType.java:57-62)
                off += 5;
            }
        } else {
            if (off + 4 > n) return fail();
            const int64_t cnt = (int32_t)(((uint32_t)data[off] << 24) | (data[off + 1] << 16) | (data[off + 2] << 8) | data[off + 3]);
            off += 4;
            if (cnt < 0 || off + 6ull * cnt > n) return fail();
            for (int64_t i = 0; i < cnt; i++) {
                const uint8_t *p = data + off;
                d->val_a.push_back((int32_t)(((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]));
                d->val_b.push_back((int32_t)(int16_t)((p[4] << 8) | p[5]));
                off += 6;
            }
        }
        d->val_off.push_back(d->val_a.size());
    }
    return (kmh_batch *)d;
}
void kmh_decoded_free(kmh_batch *h) { delete (Decoded *)h; }
uint64_t kmh_decoded_nkeys(const kmh_batch *h) { return ((const Decoded *)h)->keys.size(); }
uint64_t kmh_decoded_nvals(const kmh_batch *h) { return ((const Decoded *)h)->val_a.size(); }
const uint8_t *kmh_decoded_lens(const kmh_batch *h) { return ((const Decoded *)h)->lens.data(); }
const uint64_t *kmh_decoded_keys(const kmh_batch *h) { return ((const Decoded *)h)->keys.data(); }
const uint64_t *kmh_decoded_keys_lo(const kmh_batch *h) { return ((const Decoded *)h)->keys_lo.data(); }
const uint64_t *kmh_decoded_val_off(const kmh_batch *h) { return ((const Decoded *)h)->val_off.data(); }
const int32_t *kmh_decoded_val_a(const kmh_batch *h) { return ((const Decoded *)h)->val_a.data(); }
const int32_t *kmh_decoded_val_b(const kmh_batch *h) { return ((const Decoded *)h)->val_b.data(); }

}  // extern "C"
