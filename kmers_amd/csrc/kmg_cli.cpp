// kmg_cli.cpp -- native (C++) host side of the hot path: the three reference programs that sit on it, driven
// through the C ABI of libkmers_gpu.so, plus the on-disk container they read and write.
//
//   kmg Database.MakeDatabase -a|-q|-p -i in -o out [-K maxK -k minK] [-l keylen] [-z level | -Z] [-h] [-S maxsize] [-r readmap] [-m idmap] [-D n] [-R n] [-L a-b] [-t threads] [-P passes] [-G r/n] [-B bytes]
//   kmg Database.Matcher      -i search.db -d ref.db -o out [-Z x] [-n N [-j]] [-k a -K b] [-L a-b] [-t threads]
//   kmg Utils.TaxaCounts      -d ref.db -o out
//   kmg Utils.CollectByRead / Utils.ReadClassifier -i in -o out;  kmg Pipeline.ReadCalls -q reads.fq -d ref.db -o calls
//
// Mirrors Database/MakeDatabase.java:100-185 (option letters and defaults: -K 32 -k 24 -l 6 -z 5),
// Database/Matcher.java:60-250 (exact join, the trie search over several lengths, -n N), Utils/TaxaCounts.java:100-150
// (default mode).  The container is jar IndexedFiles.* (SURVEY.md
// Appendix A.1-A.3): header | gzip-member blocks | index block | footer; every key's block is its own gzip
// member, so blocks are deflated / inflated on a thread pool and written in key order
// (FileCreator.java:385-391 enforces that order with OrderedLatches).  Options outside the GPU path
// (k > 32, -h with minK < maxK, TaxaCounts' taxonomy modes -x / -s / -P / -N ...) exit with status 3 so that a
// wrapper can fall back to the Java program.  Plain C++17 + zlib; no HIP in this file.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/kmers_gpu.h"

extern "C" {
typedef struct kmh_batch kmh_batch;
kmh_batch *kmh_parse(const uint8_t *data, uint64_t n, int mode, int k);
void kmh_free(kmh_batch *);
uint64_t kmh_apply_id_map(kmh_batch *, const uint8_t *map, uint64_t n);
typedef struct kmh_stream kmh_stream;
kmh_stream *kmh_stream_open(int mode);
void kmh_stream_close(kmh_stream *);
kmh_batch *kmh_stream_next(kmh_stream *, const uint8_t *data, uint64_t n, int final, int k);
uint64_t kmh_nseq(const kmh_batch *);
uint64_t kmh_nbases(const kmh_batch *);
const uint8_t *kmh_bases(const kmh_batch *);
const uint64_t *kmh_seq_off(const kmh_batch *);
const int32_t *kmh_seq_id(const kmh_batch *);
const uint8_t *kmh_id_ok(const kmh_batch *);
const int64_t *kmh_seq_tok(const kmh_batch *);
uint64_t kmh_ntok(const kmh_batch *);
const uint64_t *kmh_name_off(const kmh_batch *);
const char *kmh_names(const kmh_batch *);
kmh_batch *kmh_decode_blocks(const uint8_t *data, uint64_t n, int kind);
void kmh_decoded_free(kmh_batch *);
uint64_t kmh_decoded_nkeys(const kmh_batch *);
uint64_t kmh_decoded_nvals(const kmh_batch *);
const uint64_t *kmh_decoded_keys(const kmh_batch *);
const uint64_t *kmh_decoded_val_off(const kmh_batch *);
const int32_t *kmh_decoded_val_a(const kmh_batch *);
const int32_t *kmh_decoded_val_b(const kmh_batch *);
const uint8_t *kmh_decoded_lens(const kmh_batch *);
int kmh_eof_in_kmer(const kmh_batch *);
int kmh_high_bytes(const kmh_batch *);
int kmh_empty_header(const kmh_batch *);
}

namespace {

struct Unsupported : std::runtime_error { using std::runtime_error::runtime_error; };

int g_threads = 8;

// ---------------------------------------------------------------------------------------------- helpers
std::string read_file(const std::string &path) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    std::string s;
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, n);
    fclose(f);
    return s;
}

// sequence input, gzipped or not (jar Zip.ZipOrNot: sniffed from the first two bytes; gzread does the same)
std::string read_input(const std::string &path) {
    gzFile z = gzopen(path.c_str(), "rb");
    if (!z) throw std::runtime_error("cannot open " + path);
    gzbuffer(z, 1 << 20);
    std::string s;
    std::vector<char> buf(1 << 20);
    int n;
    while ((n = gzread(z, buf.data(), (unsigned)buf.size())) > 0) s.append(buf.data(), (size_t)n);
    int errnum = Z_OK;
    (void)gzerror(z, &errnum);       // (a gzip stream that stops short: Z_BUF_ERROR after the last bytes it could hand out)
    const bool bad = n < 0 || (errnum != Z_OK && errnum != Z_STREAM_END);
    gzclose(z);
    if (bad) throw std::runtime_error("read error in " + path + " (a damaged or truncated gzip stream?)");
    return s;
}

template <typename F>
void parallel_for(size_t n, F f) {
    const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)g_threads, n));
    std::atomic<size_t> next(0);
    std::vector<std::thread> pool;
    std::atomic<bool> failed(false);
    std::string err;
    for (int t = 0; t < nt; t++)
        pool.emplace_back([&]() {
            try {
                for (size_t i; (i = next.fetch_add(1)) < n;) f(i);
            } catch (const std::exception &e) {
                if (!failed.exchange(true)) err = e.what();
            }
        });
    for (auto &th : pool) th.join();
    if (failed) throw std::runtime_error("Exception in thread: " + err);   // jar LimitedQueueExecutor.shutdownAndCheck
}

void put_be32(std::string &s, uint32_t v) { for (int i = 3; i >= 0; i--) s.push_back((char)(v >> (8 * i))); }
void put_be64(std::string &s, uint64_t v) { for (int i = 7; i >= 0; i--) s.push_back((char)(v >> (8 * i))); }
uint32_t get_be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }
uint64_t get_be64(const uint8_t *p) { return ((uint64_t)get_be32(p) << 32) | get_be32(p + 4); }

// commons-compress GzipCompressorOutputStream with GzipParameters(level): header 1f 8b 08 00 | mtime 0 |
// XFL (2 at level 9, 4 at level 1, else 0) | OS 255; raw deflate; CRC32, ISIZE  
std::string gzip_member(const uint8_t *data, size_t n, int level) {
    z_stream z;
    memset(&z, 0, sizeof z);
    if (deflateInit2(&z, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw std::runtime_error("deflateInit2");
    std::string out("\x1f\x8b\x08\x00\x00\x00\x00\x00", 8);
    out.push_back((char)(level == 9 ? 2 : (level == 1 ? 4 : 0)));
    out.push_back((char)0xFF);
    const size_t head = out.size();
    out.resize(head + deflateBound(&z, n) + 16);
    z.next_in = const_cast<Bytef *>(data);
    z.avail_in = (uInt)n;
    z.next_out = (Bytef *)&out[head];
    z.avail_out = (uInt)(out.size() - head);
    if (n > 0xFFFFFFF0ull || deflate(&z, Z_FINISH) != Z_STREAM_END) { deflateEnd(&z); throw std::runtime_error("deflate"); }
    out.resize(head + z.total_out);
    deflateEnd(&z);
    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), data, (uInt)n), isz = (uint32_t)n;
    for (int i = 0; i < 4; i++) out.push_back((char)(crc >> (8 * i)));
    for (int i = 0; i < 4; i++) out.push_back((char)(isz >> (8 * i)));
    return out;
}

// one gzip member starting at data (trailing bytes are ignored)
std::string gunzip_member(const uint8_t *data, size_t n) {
    z_stream z;
    memset(&z, 0, sizeof z);
    if (inflateInit2(&z, 15 + 16) != Z_OK) throw std::runtime_error("inflateInit2");
    std::string out;
    std::vector<uint8_t> buf(1 << 16);
    z.next_in = const_cast<Bytef *>(data);
    z.avail_in = (uInt)std::min<size_t>(n, 0xFFFFFFF0u);
    int rc;
    do {
        z.next_out = buf.data();
        z.avail_out = (uInt)buf.size();
        rc = inflate(&z, Z_NO_FLUSH);
        if (rc != Z_OK && rc != Z_STREAM_END) { inflateEnd(&z); throw std::runtime_error("corrupt gzip block"); }
        out.append((const char *)buf.data(), buf.size() - z.avail_out);
    } while (rc != Z_STREAM_END && (z.avail_in || z.avail_out == 0));
    inflateEnd(&z);
    if (rc != Z_STREAM_END) throw std::runtime_error("truncated gzip block");
    return out;
}

// ---------------------------------------------------------------------------------------------- container
const int HEADER_ZIPPED = 64, HEADER_RAW = 41;

// jar IndexedFiles.IndexedOutputFile<Integer> with an IntCompressor key (A.1).  level < 0 = unzipped (-Z).
struct IndexedOutputFile {
    FILE *f;
    int level;
    bool hr;                       // MakeDatabase -h: text index, footer flag 1
    std::map<int, std::vector<std::pair<uint64_t, uint64_t>>> index;
    uint64_t pos;

    IndexedOutputFile(const std::string &path, int zip_level, bool human = false) : level(zip_level), hr(human) {
        f = fopen(path.c_str(), "wb");
        if (!f) throw std::runtime_error("cannot create " + path);
        std::string zeros(level < 0 ? HEADER_RAW : HEADER_ZIPPED, '\0');
        fwrite(zeros.data(), 1, zeros.size(), f);
        pos = zeros.size();
    }
    std::string pack(const std::string &raw, int lvl) const {
        return level < 0 ? raw : gzip_member((const uint8_t *)raw.data(), raw.size(), lvl);
    }
    // a finished block (already deflated when the file is zipped) of `key`
    void add_block(int key, const std::string &blob) {
        if (fwrite(blob.data(), 1, blob.size(), f) != blob.size()) throw std::runtime_error("write failed");
        index[key].push_back({pos, blob.size()});
        pos += blob.size();
    }
    void close() {
        std::string idx;
        if (hr) {                  // one text line per key: key \t pos:len pos:len
            for (auto &kv : index) {
                idx += std::to_string(kv.first) + "\t";
                for (size_t i = 0; i < kv.second.size(); i++)
                    idx += (i ? " " : "") + std::to_string(kv.second[i].first) + ":" + std::to_string(kv.second[i].second);
                idx += "\n";
            }
        } else {
            put_be32(idx, (uint32_t)index.size());
            for (auto &kv : index) {
                put_be32(idx, (uint32_t)kv.first);
                put_be32(idx, (uint32_t)kv.second.size());
                for (auto &b : kv.second) { put_be64(idx, b.first); put_be64(idx, b.second); }
            }
        }
        const std::string iblob = pack(idx, level);
        const uint64_t ipos = pos;
        fwrite(iblob.data(), 1, iblob.size(), f);
        pos += iblob.size();
        char buf[128];
        snprintf(buf, sizeof buf, "%-19llu\n%-19llu\n%s\n%1d\n", (unsigned long long)ipos, (unsigned long long)iblob.size(), "1", hr ? 1 : 0);
        const std::string fblob = pack(buf, 0);          // footer and header: always level 0
        const uint64_t fpos = pos;
        fwrite(fblob.data(), 1, fblob.size(), f);
        snprintf(buf, sizeof buf, "%-19llu %-19llu \n", (unsigned long long)fpos, (unsigned long long)fblob.size());
        const std::string hblob = pack(buf, 0);
        if ((int)hblob.size() != (level < 0 ? HEADER_RAW : HEADER_ZIPPED)) throw std::runtime_error("header size");
        fseek(f, 0, SEEK_SET);
        fwrite(hblob.data(), 1, hblob.size(), f);
        if (fclose(f) != 0) throw std::runtime_error("close failed");
        f = nullptr;
    }
};

// jar IndexedFiles.IndexedInputFile<Integer>: header -> footer -> index -> blocks by key
struct IndexedInputFile {
    std::string data;
    bool zipped, hr;
    bool string_keys = false;      // key codec "2" = jar StringCompressor ([short length][bytes]): MakeDatabase -p input
    std::map<int, std::vector<std::pair<uint64_t, uint64_t>>> index;
    std::map<std::string, std::vector<std::pair<uint64_t, uint64_t>>> sindex;

    std::string block(uint64_t p, uint64_t len) const {
        if (p > data.size()) throw std::runtime_error("block outside the file");
        len = std::min<uint64_t>(len, data.size() - p);
        if (!zipped) return data.substr(p, len);
        return gunzip_member((const uint8_t *)data.data() + p, len);
    }
    explicit IndexedInputFile(const std::string &path) : data(read_file(path)) {
        zipped = data.size() >= 2 && (uint8_t)data[0] == 0x1f && (uint8_t)data[1] == 0x8b;   // Zip.ZipOrNot.isGZipped
        const std::string head = block(0, 100).substr(0, HEADER_RAW);
        unsigned long long fpos, flen, ipos, ilen;
        if (sscanf(head.c_str(), "%llu %llu", &fpos, &flen) != 2) throw std::runtime_error("bad container header: " + path);
        const std::string foot = block(fpos, flen);
        int hrflag = 0;
        char cid[32];
        if (sscanf(foot.c_str(), "%llu %llu %31s %d", &ipos, &ilen, cid, &hrflag) != 4) throw std::runtime_error("bad container footer: " + path);
        hr = hrflag == 1;
        string_keys = strcmp(cid, "2") == 0;
        const std::string raw = block(ipos, ilen);
        if (hr) {         // text index: key \t pos:len pos:len
            size_t p0 = 0;
            while (p0 < raw.size()) {
                size_t e = raw.find('\n', p0);
                if (e == std::string::npos) e = raw.size();
                const std::string line = raw.substr(p0, e - p0);
                p0 = e + 1;
                const size_t tab = line.find('\t');
                if (tab == std::string::npos) continue;
                auto &v = string_keys ? sindex[line.substr(0, tab)] : index[atoi(line.c_str())];
                const char *q = line.c_str() + tab + 1;
                while (*q) {
                    char *end;
                    const unsigned long long bp = strtoull(q, &end, 10);
                    if (*end != ':') break;
                    const unsigned long long bl = strtoull(end + 1, &end, 10);
                    v.push_back({bp, bl});
                    q = end;
                    while (*q == ' ') q++;
                }
            }
            return;
        }
        const uint8_t *p = (const uint8_t *)raw.data();
        if (raw.size() < 4) throw std::runtime_error("bad index block");
        const uint32_t n = get_be32(p);
        size_t off = 4;
        for (uint32_t i = 0; i < n; i++) {
            std::vector<std::pair<uint64_t, uint64_t>> *vp;
            uint32_t nb;
            if (string_keys) {
                if (off + 2 > raw.size()) throw std::runtime_error("bad index block");
                const size_t kl = ((size_t)p[off] << 8) | p[off + 1];
                if (off + 2 + kl + 4 > raw.size()) throw std::runtime_error("bad index block");
                vp = &sindex[std::string((const char *)p + off + 2, kl)];
                nb = get_be32(p + off + 2 + kl);
                off += 2 + kl + 4;
            } else {
                if (off + 8 > raw.size()) throw std::runtime_error("bad index block");
                vp = &index[(int)get_be32(p + off)];
                nb = get_be32(p + off + 4);
                off += 8;
            }
            auto &v = *vp;
            for (uint32_t b = 0; b < nb; b++) {
                if (off + 16 > raw.size()) throw std::runtime_error("bad index block");
                v.push_back({get_be64(p + off), get_be64(p + off + 8)});
                off += 16;
            }
        }
    }
    std::string read_key(int key) const {
        std::string out;
        auto it = index.find(key);
        if (it != index.end()) for (auto &b : it->second) out += block(b.first, b.second);
        return out;
    }
};

// path if it exists, else path.1, path.2, ... (Matcher.java:70-95,111-128)
std::vector<std::string> file_set(const std::string &path) {
    std::vector<std::string> out;
    if (FILE *f = fopen(path.c_str(), "rb")) { fclose(f); out.push_back(path); return out; }
    for (int i = 1;; i++) {
        const std::string p = path + "." + std::to_string(i);
        FILE *f = fopen(p.c_str(), "rb");
        if (!f) break;
        fclose(f);
        out.push_back(p);
    }
    return out;
}

// ---------------------------------------------------------------------------------------------- GPU side
struct Gpu {
    kmg_ctx *ctx = nullptr;
    Gpu() { if (kmg_create(0, &ctx) != KMG_OK) throw std::runtime_error("no usable GPU (kmg_create failed)"); }
    ~Gpu() { kmg_destroy(ctx); }
    void check(int rc) const {
        if (rc == KMG_OK) return;
        const std::string msg = kmg_last_error(ctx);
        if (rc == KMG_EUNSUPPORTED) throw Unsupported(msg);
        if (rc == KMG_EINVAL) throw std::runtime_error("InconsistentDataException: " + msg);
        throw std::runtime_error("kmers_gpu error " + std::to_string(rc) + ": " + msg);
    }
};

struct DbInfo { int k, key_len, rc, kind; uint64_t U, V; };
DbInfo db_info(const kmg_db *db) {
    DbInfo d;
    kmg_db_info(db, &d.k, &d.key_len, &d.rc, &d.kind, &d.U, &d.V);
    return d;
}

// FileCreator.create (FileCreator.java:117-217): meta under key -1, then every key 0..4^l-1 in order (each
// gets a block, empty ones included)
// -S maxSize (jar IndexedOutputFileSet.newFile): when the next key's block would take the file past
// maxSize, the file is closed -- `name` becomes `name.1` at the first overflow -- and `name.N+1` is opened with
// the meta block written again, so every file of the set is a complete database over a contiguous key range.
// The reference checks the limit per buffered write (it can cut inside a key's block); this writer cuts at key
// boundaries only, so the split points are equivalent, not byte-identical.
// FileCreator.create (FileCreator.java:117-217) on the output side: meta under key -1, then every key in ascending
// order, each with a block.  add() appends the blocks of keys [key_lo, key_hi) of a database; a build done in
// key-range passes calls it once per pass, in order.
struct DatabaseWriter {
    const Gpu &g;
    std::string path;
    int level;
    uint64_t max_size;
    bool hr;
    std::unique_ptr<IndexedOutputFile> out;
    std::string meta;
    int nfiles = 1;
    size_t keys_in_file = 0;
    DatabaseWriter(const Gpu &g_, const std::string &path_, int level_, uint64_t max_size_, bool hr_)
        : g(g_), path(path_), level(level_), max_size(max_size_), hr(hr_) {}
    void add(const kmg_db *db, size_t key_lo, size_t key_hi);
    void close() { if (out) out->close(); }
};

void DatabaseWriter::add(const kmg_db *db, size_t key_lo, size_t key_hi) {
    const DbInfo d = db_info(db);
    const int min_k = kmg_db_min_k(db);
    // (FileCreator.diff, FileCreator.java:234-242, runs off the end of the shorter of two k-mers where one is a prefix
    // of the other: the reference itself throws on such a database)
    if (hr && min_k != d.k) throw Unsupported("human-readable databases (-h) hold k-mers of one length on the GPU path");
    const size_t nkeys = (size_t)1 << (2 * d.key_len);
    std::vector<uint8_t> bytes;
    std::vector<uint64_t> off(nkeys + 1);
    meta.clear();
    if (!hr) {
        uint64_t nbytes = 0;
        g.check(kmg_db_encode(g.ctx, db, nullptr, 0, &nbytes, nullptr));
        bytes.resize(std::max<uint64_t>(nbytes, 1));
        g.check(kmg_db_encode(g.ctx, db, bytes.data(), nbytes, &nbytes, off.data()));
        meta.push_back((char)min_k); meta.push_back((char)d.k); meta.push_back((char)d.key_len); // FileCreator.java:153-164
        if (d.kind == KMG_PAYLOAD_TAXON) { put_be32(meta, 1); put_be32(meta, 2048); }             // CountDataType.java:115-121
        else { put_be32(meta, 2); put_be32(meta, 257); put_be32(meta, 2052); }                    // SetCompressor ++ ReadPosDataType
        meta.push_back((char)(d.rc ? 1 : 0));
    } else {
        // MakeDatabase -h (FileCreator.java:166-179, 234-242, 349-370): text meta; per key one line per k-mer,
        // the leading characters shared with the previous k-mer of the key block dropped (`last` starts as k
        // spaces), TAB, values as text: TxC|TxC with the unsaturated count, or r:p r:p
        meta = std::to_string(d.k) + "\n" + std::to_string(d.k) + "\n" + std::to_string(d.key_len) + "\n" +
               (d.kind == KMG_PAYLOAD_TAXON ? "2048" : "257 2052") + "\n" + (d.rc ? "1" : "0") + "\n";
        std::vector<uint64_t> keys(d.U + 1), val_off(d.U + 1), boff(nkeys + 1);
        std::vector<int32_t> va(d.V + 1), vb(d.V + 1);
        g.check(kmg_db_fetch(db, keys.data(), val_off.data(), va.data(), vb.data(), boff.data()));
        std::vector<std::string> text(nkeys);
        parallel_for(nkeys, [&](size_t key) {
            if (key < key_lo || key >= key_hi) return;
            std::string &t = text[key];
            std::string last((size_t)d.k, ' '), nxt((size_t)d.k, ' ');
            char buf[48];
            for (uint64_t u = boff[key]; u < boff[key + 1]; u++) {
                for (int p = 0; p < d.k; p++) nxt[p] = "ACTG"[(keys[u] >> (62 - 2 * p)) & 3];
                size_t first = 0;
                while (last[first] == nxt[first]) first++;
                t.append(nxt, first, std::string::npos);
                t.push_back('\t');
                for (uint64_t v = val_off[u]; v < val_off[u + 1]; v++) {
                    if (v != val_off[u]) t.push_back(d.kind == KMG_PAYLOAD_TAXON ? '|' : ' ');
                    snprintf(buf, sizeof buf, d.kind == KMG_PAYLOAD_TAXON ? "%dx%d" : "%d:%d", va[v], vb[v]);
                    t += buf;
                }
                t.push_back('\n');
                last = nxt;
            }
        });
        uint64_t total = 0;
        for (size_t key = 0; key < nkeys; key++) { off[key] = total; total += text[key].size(); }
        off[nkeys] = total;
        bytes.resize(std::max<uint64_t>(total, 1));
        for (size_t key = 0; key < nkeys; key++) memcpy(bytes.data() + off[key], text[key].data(), text[key].size());
    }
    std::vector<std::string> blobs(nkeys);
    if (level >= 0)
        parallel_for(nkeys, [&](size_t key) {
            if (key >= key_lo && key < key_hi) blobs[key] = gzip_member(bytes.data() + off[key], off[key + 1] - off[key], level);
        });
    if (!out) {
        out.reset(new IndexedOutputFile(path, level, hr));
        out->add_block(-1, out->pack(meta, level));
    }
    for (size_t key = key_lo; key < key_hi && key < nkeys; key++) {
        const std::string blob = level >= 0 ? std::move(blobs[key])
                                            : std::string((const char *)bytes.data() + off[key], off[key + 1] - off[key]);
        const uint64_t tail = 24 * (keys_in_file + 2) + 256;     // index + footer still to come
        if (max_size && keys_in_file > 0 && out->pos + blob.size() + tail > max_size) {
            out->close();
            if (nfiles == 1 && rename(path.c_str(), (path + ".1").c_str()) != 0) throw std::runtime_error("cannot rename " + path);
            nfiles++;
            out.reset(new IndexedOutputFile(path + "." + std::to_string(nfiles), level, hr));
            out->add_block(-1, out->pack(meta, level));
            keys_in_file = 0;
        }
        out->add_block((int)key, blob);
        keys_in_file++;
    }
}

void write_database(const Gpu &g, const kmg_db *db, const std::string &path, int level, uint64_t max_size = 0, bool hr = false) {
    DatabaseWriter w(g, path, level, max_size, hr);
    w.add(db, 0, (size_t)1 << (2 * db_info(db).key_len));
    w.close();
}

// jar Files.SizeConvertor.fromHuman: a byte count with an optional K / M / G / T suffix (powers of 1024)
uint64_t size_from_human(const std::string &v) {
    char *end = nullptr;
    double x = strtod(v.c_str(), &end);
    uint64_t mul = 1;
    if (end && *end) {
        switch (*end) {
            case 'k': case 'K': mul = 1ull << 10; break;
            case 'm': case 'M': mul = 1ull << 20; break;
            case 'g': case 'G': mul = 1ull << 30; break;
            case 't': case 'T': mul = 1ull << 40; break;
            default: throw std::runtime_error("bad size " + v);
        }
    }
    if (x <= 0) throw std::runtime_error("bad size " + v);
    return (uint64_t)(x * (double)mul);
}

// KmerFile over a human-readable file: text meta (KmerFile.java:279-283), UncompressedKmerSpliterator
// (KmerFile.java:132-160): k-mer = previous k-mer with its tail replaced by the line's first field
// (Kmer.java:26-44); values by CountDataType.fromString / SetDataType<ReadPos>.fromString
kmg_db *open_hr_database(const Gpu &g, const IndexedInputFile &f, const std::string &path, int key_lo, int key_hi) {
    const std::string m = f.read_key(-1);
    int min_k, max_k, key_len, rc;
    char ids[64];
    if (sscanf(m.c_str(), "%d\n%d\n%d\n%63[^\n]\n%d", &min_k, &max_k, &key_len, ids, &rc) != 5)
        throw std::runtime_error("bad meta block: " + path);
    if (min_k != max_k || max_k > 32) throw Unsupported("minK != maxK or k > 32: use the Java path");
    int kind;
    if (std::string(ids) == "2048") kind = KMG_PAYLOAD_TAXON;
    else if (std::string(ids) == "257 2052") kind = KMG_PAYLOAD_READPOS;
    else throw std::runtime_error("UnexpectedDataTypeException: " + path);
    std::vector<uint64_t> keys, val_off(1, 0);
    std::vector<int32_t> va, vb;
    std::string prev;
    for (auto &kv : f.index) {
        if (kv.first < 0 || kv.first < key_lo || kv.first >= key_hi) continue;
        prev.clear();                                   // (a key's first line is a whole k-mer)
        for (auto &b : kv.second) {
            const std::string text = f.block(b.first, b.second);
            size_t p0 = 0;
            while (p0 < text.size()) {
                size_t e = text.find('\n', p0);
                if (e == std::string::npos) e = text.size();
                const size_t tab = text.find('\t', p0);
                if (tab == std::string::npos || tab > e) throw std::runtime_error("bad line in " + path);
                const std::string part = text.substr(p0, tab - p0);
                const std::string cur = prev.empty() || part.size() >= prev.size() ? part : prev.substr(0, prev.size() - part.size()) + part;
                prev = cur;
                uint64_t v = 0;
                for (char c : cur) {
                    const char uc = (char)(c & 0xDF);
                    const int code = uc == 'A' ? 0 : uc == 'C' ? 1 : uc == 'T' ? 2 : uc == 'G' ? 3 : -1;
                    if (code < 0 || cur.size() > 32) throw std::runtime_error("bad k-mer in " + path);
                    v = (v << 2) | (uint64_t)code;
                }
                keys.push_back(v << (64 - 2 * cur.size()));
                const char *q = text.c_str() + tab + 1, *qe = text.c_str() + e;
                while (q < qe) {
                    char *end;
                    const long long a = strtoll(q, &end, 10);
                    const long long bb = strtoll(end + 1, &end, 10);          // after 'x' or ':'
                    va.push_back((int32_t)a);
                    vb.push_back((int32_t)std::min<long long>(bb, 0x7FFFFFFF));
                    q = end;
                    if (q < qe) q++;                                            // '|' or ' '
                }
                val_off.push_back(va.size());
                p0 = e + 1;
            }
        }
    }
    kmg_db *db = nullptr;
    keys.push_back(0); va.push_back(0); vb.push_back(0);   // never null
    g.check(kmg_db_load(g.ctx, max_k, key_len, rc, kind, keys.data(), keys.size() - 1, val_off.data(), va.data(), vb.data(), &db));
    return db;
}

// new KmerFile(file, dataType) + its decoded content uploaded to the GPU
// keys [key_lo, key_hi) only: Matcher -L
kmg_db *load_database(const Gpu &g, const IndexedInputFile &f, const std::string &path, int key_lo = 0, int key_hi = INT32_MAX) {
    if (f.hr) return open_hr_database(g, f, path, key_lo, key_hi);
    const std::string m = f.read_key(-1);                                   // KmerFile.getMetaData, KmerFile.java:284-288
    if (m.size() < 8) throw std::runtime_error("bad meta block: " + path);
    const int min_k = (uint8_t)m[0], max_k = (uint8_t)m[1], key_len = (uint8_t)m[2];
    const uint32_t nid = get_be32((const uint8_t *)m.data() + 3);
    if (m.size() < 8 + 4ull * nid) throw std::runtime_error("bad meta block: " + path);
    std::vector<uint32_t> ids(nid);
    for (uint32_t i = 0; i < nid; i++) ids[i] = get_be32((const uint8_t *)m.data() + 7 + 4 * i);
    const int rc = m[7 + 4 * nid] == 1;
    if (max_k > 32) throw Unsupported("k > 32: use the Java path");
    int kind;
    if (ids == std::vector<uint32_t>{2048}) kind = KMG_PAYLOAD_TAXON;
    else if (ids == std::vector<uint32_t>{257, 2052}) kind = KMG_PAYLOAD_READPOS;
    else throw std::runtime_error("UnexpectedDataTypeException: " + path);   // KmerFile.java:30-33
    // Every key's block is a stream of its own (its first record shares nothing, FileCreator.java:315), so the keys
    // are inflated and decoded side by side and their arrays laid end to end in key order.
    std::vector<const std::vector<std::pair<uint64_t, uint64_t>> *> per_key;
    for (auto &kv : f.index) if (kv.first >= 0 && kv.first >= key_lo && kv.first < key_hi) per_key.push_back(&kv.second);
    const size_t nk = per_key.size();
    std::vector<kmh_batch *> dec(nk, nullptr);
    std::atomic<bool> bad{false};
    parallel_for(nk, [&](size_t i) {
        std::string bytes;
        for (auto &b : *per_key[i]) bytes += f.block(b.first, b.second);
        dec[i] = kmh_decode_blocks((const uint8_t *)bytes.data(), bytes.size(), kind);
        if (!dec[i]) bad = true;
    });
    std::vector<uint64_t> kbase(nk + 1, 0), vbase(nk + 1, 0);
    for (size_t i = 0; i < nk && !bad; i++) {
        kbase[i + 1] = kbase[i] + kmh_decoded_nkeys(dec[i]);
        vbase[i + 1] = vbase[i] + kmh_decoded_nvals(dec[i]);
    }
    std::vector<uint64_t> keys, val_off;
    std::vector<int32_t> va, vb;
    std::vector<uint8_t> lens;
    const bool var = min_k != max_k;
    if (!bad) {
        keys.resize(kbase[nk] + 1); val_off.resize(kbase[nk] + 1); va.resize(vbase[nk] + 1); vb.resize(vbase[nk] + 1);
        if (var) lens.resize(kbase[nk] + 1);
        parallel_for(nk, [&](size_t i) {
            const uint64_t n = kmh_decoded_nkeys(dec[i]), v = kmh_decoded_nvals(dec[i]);
            if (n) memcpy(&keys[kbase[i]], kmh_decoded_keys(dec[i]), n * 8);
            if (n && var) memcpy(&lens[kbase[i]], kmh_decoded_lens(dec[i]), n);
            const uint64_t *o = kmh_decoded_val_off(dec[i]);
            for (uint64_t j = 0; j < n; j++) val_off[kbase[i] + j] = vbase[i] + o[j];
            if (v) {
                memcpy(&va[vbase[i]], kmh_decoded_val_a(dec[i]), v * 4);
                memcpy(&vb[vbase[i]], kmh_decoded_val_b(dec[i]), v * 4);
            }
        });
        val_off[kbase[nk]] = vbase[nk];
    }
    for (auto d : dec) if (d) kmh_decoded_free(d);
    if (bad) throw std::runtime_error("corrupt k-mer blocks (or k > 32): " + path);
    kmg_db *db = nullptr;
    if (var)
        g.check(kmg_db_load_var(g.ctx, max_k, min_k, key_len, rc, kind, keys.data(), lens.data(), kbase[nk], val_off.data(), va.data(),
                                vb.data(), &db));
    else
        g.check(kmg_db_load(g.ctx, max_k, key_len, rc, kind, keys.data(), kbase[nk], val_off.data(), va.data(), vb.data(), &db));
    return db;
}

// a database given as one file or as name.1 .. name.N (Matcher.java:80-89), opened once; load() decodes the keys
// [key_lo, key_hi) of every file and merges the files on the GPU (DB.kmers, DB.java:99-109)
struct DatabaseSet {
    std::vector<std::string> paths;
    std::vector<std::unique_ptr<IndexedInputFile>> files;
    uint64_t bytes = 0;
    int end_key = 0;              // one past the largest key any file holds
    explicit DatabaseSet(const std::string &path) : paths(file_set(path)) {
        if (paths.empty()) throw std::runtime_error("cannot open " + path);
        for (auto &p : paths) {
            files.emplace_back(new IndexedInputFile(p));
            bytes += files.back()->data.size();
            if (!files.back()->index.empty()) end_key = std::max(end_key, files.back()->index.rbegin()->first + 1);
        }
    }
    kmg_db *load(const Gpu &g, int key_lo = 0, int key_hi = INT32_MAX) const;
};

kmg_db *open_database_set(const Gpu &g, const std::string &path, int key_lo = 0, int key_hi = INT32_MAX) {
    return DatabaseSet(path).load(g, key_lo, key_hi);
}

kmg_db *DatabaseSet::load(const Gpu &g, int key_lo, int key_hi) const {
    std::vector<kmg_db *> dbs;
    for (size_t i = 0; i < files.size(); i++) dbs.push_back(load_database(g, *files[i], paths[i], key_lo, key_hi));
    if (dbs.size() == 1) return dbs[0];
    kmg_db *out = nullptr;
    const int r = kmg_db_merge(g.ctx, dbs.data(), (int)dbs.size(), &out);
    for (auto *d : dbs) kmg_db_free(d);
    g.check(r);
    return out;
}

// ---------------------------------------------------------------------------------------------- programs
struct Args {
    std::map<char, std::string> val;
    std::map<char, bool> flag;
    bool has(char c) const { return val.count(c) || flag.count(c); }
    std::string get(char c, const std::string &dflt = "") const { auto it = val.find(c); return it == val.end() ? dflt : it->second; }
    int geti(char c, int dflt) const { auto it = val.find(c); return it == val.end() ? dflt : atoi(it->second.c_str()); }
};
Args parse_args(int argc, char **argv, int from, const std::string &with_value, const std::string &flags) {
    Args a;
    for (int i = from; i < argc; i++) {
        const std::string s = argv[i];
        if (s.size() != 2 || s[0] != '-') throw std::runtime_error("unexpected argument " + s);
        if (flags.find(s[1]) != std::string::npos) a.flag[s[1]] = true;
        else if (with_value.find(s[1]) != std::string::npos) {
            if (i + 1 >= argc) throw std::runtime_error("option " + s + " needs a value");
            a.val[s[1]] = argv[++i];
        } else throw Unsupported("option " + s + " is not on the GPU path");
    }
    return a;
}

// -v: wall-clock of every stage on stderr
struct StageClock {
    bool on;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    explicit StageClock(bool v) : on(v) {}
    void lap(const char *what) {
        const auto n = std::chrono::steady_clock::now();
        if (on) fprintf(stderr, "  %-28s %8.3f s\n", what, std::chrono::duration<double>(n - t).count());
        t = n;
    }
};

// MakeDatabase -p (MakeDatabase.java:186-237,320-420): the input is the IndexedFile OtherFiles.SeqToTaxID writes --
// String keys, blocks of (taxid, Sequence) records, binary [int taxid][int length][2 bits per base] or
// human-readable lines `taxid ~ SEQUENCE` -- and every window of every sequence goes into one reference database.
int make_database_preprocessed(const Args &a, int k, int min_k, int key_len, int level) {
    StageClock clk(a.has('v'));
    IndexedInputFile in(a.get('i'));
    if (!in.string_keys) throw std::runtime_error("not a preprocessed file (its keys are not strings): " + a.get('i'));
    std::string bases;                 // letters (human-readable input) or packed bytes (binary input)
    std::vector<uint64_t> off{0}, boff{0};
    std::vector<int32_t> ids;
    // -L start[-end]: only the input indexes between the two names, both inclusive (MakeDatabase.java:207-227)
    std::string l_start, l_end;
    if (a.has('L')) {
        const std::string L = a.get('L');
        const size_t dash = L.find('-');
        l_start = L.substr(0, dash);
        l_end = dash == std::string::npos ? l_start : L.substr(dash + 1);
    }
    for (auto &kv : in.sindex) {
        if (a.has('L') && (kv.first.compare(l_start) < 0 || kv.first.compare(l_end) > 0)) continue;
        std::string data;
        for (auto &b : kv.second) data += in.block(b.first, b.second);
        if (in.hr) {
            size_t p0 = 0;
            while (p0 < data.size()) {
                size_t e = data.find('\n', p0);
                if (e == std::string::npos) e = data.size();
                const size_t sep = data.find(" ~ ", p0);
                if (e > p0) {
                    if (sep == std::string::npos || sep >= e) throw std::runtime_error("bad preprocessed line");
                    ids.push_back((int32_t)strtol(data.c_str() + p0, nullptr, 10));
                    bases.append(data, sep + 3, e - sep - 3);
                    off.push_back(bases.size());
                }
                p0 = e + 1;
            }
        } else {
            const uint8_t *p = (const uint8_t *)data.data();
            size_t o = 0;
            while (o + 8 <= data.size()) {
                const int32_t t = (int32_t)get_be32(p + o);
                const uint32_t n = get_be32(p + o + 4);
                const size_t nb = n ? (n - 1) / 4 + 1 : 0;
                if (o + 8 + nb > data.size()) throw std::runtime_error("truncated preprocessed record");
                bases.append((const char *)p + o + 8, nb);       // stays packed: the device restores the letters
                ids.push_back(t);
                off.push_back(off.back() + n);
                boff.push_back(bases.size());
                o += 8 + nb;
            }
        }
    }
    clk.lap("read + unpack input");
    Gpu g;
    g.check(kmg_build_begin(g.ctx, k, key_len, 1, KMG_PAYLOAD_TAXON));
    if (a.has('D') || a.has('R')) g.check(kmg_build_set_filters(g.ctx, a.geti('D', -1), a.geti('R', -1)));
    // PreProcessedSpliterator with minK < maxK (MakeDatabase.java:393-404) hands out from every start the k-mer of
    // min(maxK, bases left) bases: every window plus the suffixes of length maxK-1 .. minK at the sequence's end --
    // what the text path yields for the same, all-valid, sequence
    if (min_k != k) g.check(kmg_build_set_min_k(g.ctx, min_k));
    if (!ids.empty()) {
        if (in.hr)
            g.check(kmg_build_add(g.ctx, (const uint8_t *)bases.data(), bases.size(), off.data(), ids.data(), nullptr, ids.size(), nullptr));
        else
            g.check(kmg_build_add_packed(g.ctx, (const uint8_t *)bases.data(), bases.size(), off.data(), boff.data(), ids.data(), ids.size()));
    }
    kmg_db *db = nullptr;
    uint64_t nwin = 0;
    g.check(kmg_build_finish(g.ctx, &db, &nwin));
    clk.lap("copy in + build (GPU)");
    write_database(g, db, a.get('o'), level, a.has('S') ? size_from_human(a.get('S')) : 0, a.has('h'));
    clk.lap("encode + deflate + write");
    kmg_db_free(db);
    return 0;
}

// A sequence file read, inflated and split into records piece by piece on a thread of its own (kmh_stream_*): the
// consumer takes the pieces' batches in order while the next ones are being made.  Neither the text nor the sequences
// of the whole input are in memory at any time; the reader runs ahead by at most ~256 MiB of input -- plus the record it
// is in the middle of: a record waits whole (as text, appended piece by piece without being parsed again) until the next
// header start shows, so a chromosome-sized FASTA record is in memory once.
struct Piece {
    std::unique_ptr<kmh_batch, void (*)(kmh_batch *)> batch{nullptr, kmh_free};
    bool last = false;
    std::string err;
};
class PieceReader {
  public:
    // mode: 0 FASTA, 1 FASTQ (kmh_parse's); k: for the FASTA repeated-last-k-mer rule at the end of the input
    PieceReader(const std::string &path, int mode, int k, uint64_t piece_bytes)
        : path_(path), mode_(mode), k_(k), piece_bytes_(std::max<uint64_t>(1, piece_bytes)),
          cap_((size_t)std::max<uint64_t>(2, (256ull << 20) / std::max<uint64_t>(1, piece_bytes))), th_([this] { run(); }) {}
    ~PieceReader() {      // whatever ends the consumer, the reader is told and joined
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        if (th_.joinable()) th_.join();
    }
    Piece next() {
        Piece p;
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return !ready_.empty(); });
            p = std::move(ready_.front());
            ready_.pop_front();
        }
        cv_.notify_all();
        if (!p.err.empty()) throw std::runtime_error(p.err);
        return p;
    }

  private:
    void push(Piece &&p) {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || ready_.size() < cap_; });
        if (!stop_) ready_.push_back(std::move(p));
        cv_.notify_all();
    }
    void fill(gzFile z, std::vector<uint8_t> &buf) {      // up to piece_bytes, grown as the file delivers
        buf.clear();
        uint64_t got = 0;
        while (got < piece_bytes_) {
            const uint64_t want = std::min<uint64_t>(piece_bytes_ - got, 16u << 20);
            buf.resize(got + want);
            const int n = gzread(z, buf.data() + got, (unsigned)want);
            if (n < 0) throw std::runtime_error("read error (a damaged gzip stream?)");
            if (n == 0) {
                // (a gzip stream that stops short hands out what it has and reports Z_BUF_ERROR: the reference's
                // GZIPInputStream throws EOFException there)
                int errnum = Z_OK;
                (void)gzerror(z, &errnum);
                if (errnum != Z_OK && errnum != Z_STREAM_END) throw std::runtime_error("unexpected end of the gzip stream");
                break;
            }
            got += (uint64_t)n;
        }
        buf.resize(got);
    }
    void run() {
        kmh_stream *st = kmh_stream_open(mode_);
        gzFile z = gzopen(path_.c_str(), "rb");      // (jar Zip.ZipOrNot: gzip or not, sniffed; gzread does the same)
        if (!z) {
            Piece p; p.last = true; p.err = "cannot open " + path_;
            push(std::move(p));
        } else {
            gzbuffer(z, 1 << 20);
            try {
                std::vector<uint8_t> cur, nxt;
                fill(z, cur);
                for (;;) {
                    fill(z, nxt);                            // (what follows decides whether `cur` is the last piece)
                    Piece p;
                    p.last = nxt.empty();
                    p.batch.reset(kmh_stream_next(st, cur.data(), cur.size(), p.last ? 1 : 0, k_));
                    const bool last = p.last;
                    push(std::move(p));
                    { std::lock_guard<std::mutex> lk(mu_); if (stop_) break; }
                    if (last) break;
                    cur.swap(nxt);
                }
            } catch (const std::exception &e) {
                Piece p; p.last = true; p.err = "reading " + path_ + ": " + e.what();
                push(std::move(p));
            }
            gzclose(z);
        }
        kmh_stream_close(st);
    }
    const std::string path_;
    const int mode_, k_;
    const uint64_t piece_bytes_;
    const size_t cap_;
    std::deque<Piece> ready_;
    std::mutex mu_;
    std::condition_variable cv_;
    bool stop_ = false;
    std::thread th_;      // (last: it starts in the constructor and uses the members above)
};

// FASTA (reference) or FASTQ (reads) text -> database on the GPU; for reads also the read-name map file
// (ReadIDMapping.geNext prints a header each time a new index is handed out: gzip text, one per line) unless
// read_map is empty, and the number of read indices.
// passes > 1: the database is built in that many key ranges, one after the other over the same parsed input
// (kmg_build_set_key_range), each handed to per_pass(db, first key, end key) and freed; returns NULL then.
kmg_db *build_from_text(Gpu &g, const std::string &path, bool reads, int k, int key_len, bool filters, int dust, int runofsame,
                        const std::string &read_map, uint64_t *n_reads, uint64_t *nwin, StageClock &clk, int passes = 1,
                        const std::function<void(kmg_db *, size_t, size_t)> &per_pass = nullptr, int share = 0, int shares = 1,
                        int min_k = 0, const std::string &id_map = "", uint64_t piece_bytes = 16ull << 20) {
    if (min_k <= 0) min_k = k;
    // id_map (-m, FASTA only; MakeDatabase.java:138-147): headers are names looked up in the map file, not ints
    const bool named = !reads && !id_map.empty();
    if (!named) {
        // The input in pieces (PieceReader): piece i + 1 is read, inflated and split while piece i is copied in and its
        // k-mers extracted; one kmg_build_add per piece, the read numbering runs on from add to add.  A build in key
        // ranges (-P) reads the file once per range.
        const size_t nkeys = (size_t)1 << (2 * key_len);
        const size_t r_lo = nkeys * (size_t)share / (size_t)shares, r_hi = nkeys * ((size_t)share + 1) / (size_t)shares;
        const bool ranged = passes > 1 || shares > 1;
        passes = (int)std::max<size_t>(1, std::min<size_t>((size_t)passes, std::max<size_t>(r_hi - r_lo, 1)));
        uint64_t windows = 0;
        kmg_db *db = nullptr;
        int32_t top = -1;
        bool map_done = false;
        for (int ps = 0; ps < passes; ps++) {
            const size_t lo = r_lo + (r_hi - r_lo) * ps / passes, hi = r_lo + (r_hi - r_lo) * (ps + 1) / passes;
            if (hi <= lo) continue;
            PieceReader rd(path, reads ? 1 : 0, k, piece_bytes);
            g.check(kmg_build_begin(g.ctx, k, key_len, reads ? 0 : 1, reads ? KMG_PAYLOAD_READPOS : KMG_PAYLOAD_TAXON));
            if (filters) g.check(kmg_build_set_filters(g.ctx, dust, runofsame));
            if (min_k != k) g.check(kmg_build_set_min_k(g.ctx, min_k));
            if (ranged) g.check(kmg_build_set_key_range(g.ctx, (uint32_t)lo, (uint32_t)hi));
            gzFile z = nullptr;
            if (reads && !read_map.empty() && !map_done) {
                z = gzopen(read_map.c_str(), "wb");
                if (!z) throw std::runtime_error("cannot create " + read_map);
            }
            struct CloseZ { gzFile &z; ~CloseZ() { if (z) gzclose(z); } } close_z{z};
            int32_t last = -1;
            uint64_t seq_base = 0;
            for (bool done = false; !done;) {
                Piece pc = rd.next();
                const kmh_batch *b = pc.batch.get();
                const uint64_t nseq = kmh_nseq(b);
                const uint64_t *off = kmh_seq_off(b);
                // Sequence.java:15-33 decides by the sign of a window's first byte: windows made of bytes >= 128 only are
                // k-mers of garbage codes there, invalid here -- refused rather than built differently
                if (kmh_high_bytes(b)) throw Unsupported("sequence lines hold bytes >= 128: not supported on the GPU path");
                // ReadIDMapping starts with curReadS = "" (ReadIDMapping.java:10-16): a first emitting read with an empty
                // header gets index -1 and no name line there
                if (reads && kmh_empty_header(b)) throw Unsupported("a FASTQ record with an empty header: not supported on the GPU path");
                // A file that ends inside a sequence line: the reference hands its last k-mer out twice
                // (KmersFromFile.java:115-123,193-200).  For FASTA with k-mers of one length the record splitter reproduces
                // that; a FASTQ cut there, or any such file with minK < maxK, is refused rather than built differently.
                if (pc.last && kmh_eof_in_kmer(b) && (reads || min_k != k))
                    throw Unsupported("input ends inside a sequence line (no final newline): not supported on the GPU path for FASTQ input or with minK < maxK");
                // Integer.parseInt (or the -m map) throws at a sequence's first k-mer and not before: new Kmer(..) is evaluated
                // first, KmersFromFile.java:103,165.  A sequence whose header is no int is handed over with whatever taxon
                // the splitter left there, and the build says which sequences produced a k-mer.
                const uint8_t *ok = reads ? nullptr : kmh_id_ok(b);
                bool any_bad = false;
                for (uint64_t q = 0; ok && q < nseq && !any_bad; q++) any_bad = !ok[q];
                std::vector<int32_t> ridx(reads || any_bad ? nseq : 0);
                if (nseq)
                    g.check(kmg_build_add(g.ctx, kmh_bases(b), kmh_nbases(b), off, reads ? nullptr : kmh_seq_id(b),
                                          reads ? kmh_seq_tok(b) : nullptr, nseq, ridx.empty() ? nullptr : ridx.data()));
                for (uint64_t q = 0; any_bad && q < nseq; q++)
                    if (!ok[q] && ridx[q] >= 0)
                        throw std::runtime_error("FASTA header of sequence " + std::to_string(seq_base + q) + (named ? " is not in the id map" : " is not an int"));
                if (reads) {
                    const uint64_t *noff = kmh_name_off(b);      // (a stream's piece: one name per sequence)
                    const char *names = kmh_names(b);
                    for (uint64_t q = 0; q < nseq; q++) {
                        top = std::max(top, ridx[q]);
                        if (z && ridx[q] >= 0 && ridx[q] != last) {
                            gzwrite(z, names + noff[q], (unsigned)(noff[q + 1] - noff[q]));
                            gzwrite(z, "\n", 1);
                            last = ridx[q];
                        }
                    }
                }
                seq_base += nseq;
                done = pc.last;
            }
            if (z) map_done = true;
            uint64_t w = 0;
            g.check(kmg_build_finish(g.ctx, &db, &w));
            windows += w;
            if (ranged) {
                per_pass(db, lo, hi);
                kmg_db_free(db);
                db = nullptr;
            }
        }
        if (nwin) *nwin = windows;
        if (reads && n_reads) *n_reads = (uint64_t)(top + 1);
        clk.lap(ranged ? "read + split alongside copy in + build (GPU), encode + write, per pass" : "read + split alongside copy in + build (GPU)");
        return db;
    }
    const std::string data = read_input(path);
    clk.lap("read input");
    const std::unique_ptr<kmh_batch, void (*)(kmh_batch *)> batch(kmh_parse((const uint8_t *)data.data(), data.size(), reads ? 1 : (named ? 2 : 0), k), kmh_free);
    if (named) {
        const std::string map = read_input(id_map);
        const uint64_t bad = kmh_apply_id_map(batch.get(), (const uint8_t *)map.data(), map.size());
        if (bad) throw std::runtime_error("line " + std::to_string(bad) + " of the id map is not <sequence id><tab><int>");
    }
    const kmh_batch *b = batch.get();
    const uint64_t nseq = kmh_nseq(b);
    const uint64_t *off = kmh_seq_off(b);
    // A file that ends inside a sequence line: the reference hands its last k-mer out twice (KmersFromFile.java:115-123,
    // 193-200).  For FASTA with k-mers of one length the record splitter reproduces that; a FASTQ cut there, or any such
    // file with minK < maxK (where suffixes follow the repeated k-mer), is refused rather than built differently.
    // Sequence.java:15-33 decides by the sign of a window's first byte: windows made of bytes >= 128 only are k-mers of
    // garbage codes there, invalid here -- refused rather than built differently
    if (kmh_high_bytes(b)) throw Unsupported("sequence lines hold bytes >= 128: not supported on the GPU path");
    // ReadIDMapping starts with curReadS = "" (ReadIDMapping.java:10-16): a first emitting read with an empty header gets
    // index -1 and no name line there
    if (reads && kmh_empty_header(b)) throw Unsupported("a FASTQ record with an empty header: not supported on the GPU path");
    if (kmh_eof_in_kmer(b) && (reads || min_k != k))
        throw Unsupported("input ends inside a sequence line (no final newline): not supported on the GPU path for FASTQ input or with minK < maxK");
    // Integer.parseInt (or the -m map) throws at a sequence's first k-mer and not before (new Kmer(..) is evaluated first,
    // KmersFromFile.java:103,165): the build says which sequences produced one
    const uint8_t *id_ok = reads ? nullptr : kmh_id_ok(b);
    bool any_bad = false;
    for (uint64_t s = 0; id_ok && s < nseq && !any_bad; s++) any_bad = !id_ok[s];
    clk.lap("split records (host)");
    std::vector<int32_t> ridx(reads || any_bad ? nseq : 0);
    kmg_db *db = nullptr;
    const size_t nkeys = (size_t)1 << (2 * key_len);
    // share / shares (-G): this process builds the share-th of `shares` equal key ranges only
    const size_t r_lo = nkeys * (size_t)share / (size_t)shares, r_hi = nkeys * ((size_t)share + 1) / (size_t)shares;
    const bool ranged = passes > 1 || shares > 1;
    passes = (int)std::max<size_t>(1, std::min<size_t>((size_t)passes, std::max<size_t>(r_hi - r_lo, 1)));
    uint64_t windows = 0;
    for (int ps = 0; ps < passes; ps++) {
        const size_t lo = r_lo + (r_hi - r_lo) * ps / passes, hi = r_lo + (r_hi - r_lo) * (ps + 1) / passes;
        if (hi <= lo) continue;
        g.check(kmg_build_begin(g.ctx, k, key_len, reads ? 0 : 1, reads ? KMG_PAYLOAD_READPOS : KMG_PAYLOAD_TAXON));
        if (filters) g.check(kmg_build_set_filters(g.ctx, dust, runofsame));
        if (min_k != k) g.check(kmg_build_set_min_k(g.ctx, min_k));
        if (ranged) g.check(kmg_build_set_key_range(g.ctx, (uint32_t)lo, (uint32_t)hi));
        g.check(kmg_build_add(g.ctx, kmh_bases(b), kmh_nbases(b), off, reads ? nullptr : kmh_seq_id(b),
                              reads ? kmh_seq_tok(b) : nullptr, nseq, ridx.empty() ? nullptr : ridx.data()));
        for (uint64_t s = 0; any_bad && s < nseq; s++)
            if (!id_ok[s] && ridx[s] >= 0)
                throw std::runtime_error("FASTA header of sequence " + std::to_string(s) + (named ? " is not in the id map" : " is not an int"));
        uint64_t w = 0;
        g.check(kmg_build_finish(g.ctx, &db, &w));
        windows += w;
        if (ranged) {
            per_pass(db, lo, hi);
            kmg_db_free(db);
            db = nullptr;
        }
    }
    if (nwin) *nwin = windows;
    clk.lap(ranged ? "copy in + build (GPU), encode + write, per pass" : "copy in + build (GPU)");
    if (reads) {
        int32_t top = -1;
        for (uint64_t s = 0; s < nseq; s++) top = std::max(top, ridx[s]);
        if (n_reads) *n_reads = (uint64_t)(top + 1);
        if (!read_map.empty()) {
            gzFile z = gzopen(read_map.c_str(), "wb");
            if (!z) { if (db) kmg_db_free(db); throw std::runtime_error("cannot create " + read_map); }
            const int64_t *tok = kmh_seq_tok(b);
            const uint64_t *noff = kmh_name_off(b);
            const char *names = kmh_names(b);
            int32_t last = -1;
            for (uint64_t s = 0; s < nseq; s++)
                if (ridx[s] >= 0 && ridx[s] != last) {
                    gzwrite(z, names + noff[tok[s]], (unsigned)(noff[tok[s] + 1] - noff[tok[s]]));
                    gzwrite(z, "\n", 1);
                    last = ridx[s];
                }
            gzclose(z);
            clk.lap("read-name map");
        }
    }
    return db;
}

int make_database(int argc, char **argv) {
    const Args a = parse_args(argc, argv, 2, "ioKklzrDRtcSmNPLGB", "aqpZvh");
    if (a.has('L') && !a.has('p')) throw std::runtime_error("-L limits the input indexes of a preprocessed file (-p)");
    for (char c : std::string("N")) if (a.has(c)) throw Unsupported(std::string("option -") + c + " is not on the GPU path");
    const int k = a.geti('K', 32), min_k = a.geti('k', 24), key_len = a.geti('l', 6);     // MakeDatabase.java:106-109
    const int level = a.has('Z') ? -1 : a.geti('z', 5);
    g_threads = std::max(1, a.geti('t', g_threads));
    if (min_k > k) throw std::runtime_error("-k (shortest k-mer) above -K (longest)");
    if (min_k != k && a.has('h')) throw Unsupported("human-readable databases (-h) hold k-mers of one length on the GPU path");
    // (k-mers of 33 .. 64 bases: the library builds and encodes them; this driver's text writer knows one word per k-mer)
    if (k > 32 && a.has('h')) throw Unsupported("human-readable databases (-h) of k-mers of more than 32 bases: use the Java path");
    if ((int)a.has('a') + (int)a.has('q') + (int)a.has('p') != 1)
        throw std::runtime_error("exactly one of -a (FASTA reference) / -q (FASTQ reads) / -p (preprocessed) is required");
    if (!a.has('i') || !a.has('o')) throw std::runtime_error("-i and -o are required");
    if (a.has('p') && (a.has('P') || a.has('G'))) throw std::runtime_error("-P and -G go with -a / -q");
    if (a.has('p')) return make_database_preprocessed(a, k, min_k, key_len, level);
    const bool reads = a.has('q');
    if (reads && !a.has('r')) throw std::runtime_error("-r <read map file> is required with -q");   // MakeDatabase.java:172-173
    StageClock clk(a.has('v'));
    Gpu g;
    clk.lap("GPU context");
    uint64_t nwin = 0;
    // -P n (no counterpart in the reference): build and write the database in n key ranges, one after the other --
    // the same file, for inputs whose k-mers do not fit one build (2^32-16 records) or the GPU's memory
    const int passes = std::max(1, a.geti('P', 1));
    // -G r/n (no counterpart either): this process builds the r-th of n equal key ranges (r = 0 .. n-1) and writes it as
    // <output>.<r+1> -- n processes, one per GPU, no communication, and the files are a database set as the reference
    // reads it (name.1 .. name.N, Matcher.java:80-89).  The read map is written by share 0 only.
    int share = 0, shares = 1;
    if (a.has('G')) {
        if (sscanf(a.get('G').c_str(), "%d/%d", &share, &shares) != 2 || shares < 1 || share < 0 || share >= shares)
            throw std::runtime_error("-G wants r/n with 0 <= r < n");
        if (a.has('S')) throw std::runtime_error("-G and -S both name the files of a set");
    }
    const std::string out_path = shares > 1 ? a.get('o') + "." + std::to_string(share + 1) : a.get('o');
    DatabaseWriter w(g, out_path, level, a.has('S') ? size_from_human(a.get('S')) : 0, a.has('h'));
    uint64_t U = 0, V = 0;
    auto emit = [&](kmg_db *part, size_t lo, size_t hi) {
        const DbInfo d = db_info(part);
        U += d.U; V += d.V;
        w.add(part, lo, hi);
    };
    kmg_db *db = build_from_text(g, a.get('i'), reads, k, key_len, a.has('D') || a.has('R'), a.geti('D', -1), a.geti('R', -1),
                                 reads && share == 0 ? a.get('r') : "", nullptr, &nwin, clk, passes, emit, share, shares, min_k,
                                 a.has('m') && !reads ? a.get('m') : "",
                                 a.has('B') ? size_from_human(a.get('B')) : (16ull << 20));    // -B: bytes of input per piece
    if (db) {
        emit(db, 0, (size_t)1 << (2 * key_len));
        kmg_db_free(db);
    }
    w.close();
    clk.lap("encode + deflate + write");
    if (a.has('v'))
        fprintf(stderr, "%llu k-mer windows, %llu distinct k-mers, %llu values\n", (unsigned long long)nwin,
                (unsigned long long)U, (unsigned long long)V);
    return 0;
}

struct HostDb {
    DbInfo d;
    std::vector<uint64_t> keys, val_off;
    std::vector<int32_t> a, b;
};
HostDb fetch(const Gpu &g, const kmg_db *db) {
    HostDb h;
    h.d = db_info(db);
    h.keys.resize(h.d.U + 1); h.val_off.resize(h.d.U + 1); h.a.resize(h.d.V + 1); h.b.resize(h.d.V + 1);
    g.check(kmg_db_fetch(db, h.keys.data(), h.val_off.data(), h.a.data(), h.b.data(), nullptr));
    return h;
}

// "r:p|r:p" (SetCompressor.toString over ReadPosDataType.java:46-49) or "TxC|TxC" (CountDataType.java:109-113;
// counts are what the files hold -- bytes saturated at 255 when written, CountDataType.java:40 -- and, for a database
// given as several files, their sums: DB.kmers adds them as longs without saturating again, DB.java:99-109, so a line
// can show more than 255)
// decimal text of a Java int, appended (the match text holds hundreds of millions of them)
inline void append_int(std::string &s, int32_t v) {
    char buf[12];
    char *e = buf + sizeof buf, *p = e;
    uint32_t u = v < 0 ? 0u - (uint32_t)v : (uint32_t)v;
    do { *--p = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) *--p = '-';
    s.append(p, (size_t)(e - p));
}
void append_values(std::string &s, const HostDb &h, uint64_t i) {
    const bool reads = h.d.kind == KMG_PAYLOAD_READPOS;
    for (uint64_t v = h.val_off[i]; v < h.val_off[i + 1]; v++) {
        if (v != h.val_off[i]) s.push_back('|');
        append_int(s, h.a[v]);
        s.push_back(reads ? ':' : 'x');
        append_int(s, h.b[v]);
    }
}

// Kmer.getRC on a left-aligned 2-bit k-mer
uint64_t revcomp_key(uint64_t key, int k) {
    uint64_t v = key >> (64 - 2 * k), out = 0;
    for (int i = 0; i < k; i++) { out = (out << 2) | ((v & 3) ^ 2); v >>= 2; }
    return out << (64 - 2 * k);
}

// The third field of an inexact match line: `DIFF|TxC|TxC DIFF|TxC ...` from the matches of search k-mer i, which
// arrive in ascending reference order.  -j keeps the best distance only (Root.java:144-148, DB.java:164-171); of a
// k-mer and its reverse complement the one met first stays unless the later one is closer
// (ClosestInfoCollector.add, ClosestInfoCollector.java:30-50); entries are ordered by number of differences, then by
// (position, base) in turn (KmerDiffComparaotr, :119-147); a difference prints as position + new base, none as "-"
// (KmerDiff.java:54-67).
// vals / row_of: the values printed for reference k-mer h are row row_of[h] of `vals` instead of h's own (the search over
// several lengths: what the trie holds for a prefix, kmg_match_trie)
void append_near(std::string &s, const HostDb &hs, const HostDb &hr, uint64_t i, int k, const std::vector<uint64_t> &off,
                 const std::vector<int64_t> &ref, const std::vector<uint8_t> &dist, bool just, const HostDb *vals = nullptr,
                 const std::vector<int64_t> *row_of = nullptr) {
    struct Kept { uint64_t key; int64_t h; int d; bool alive; };
    std::vector<Kept> kept;
    int best = 255;
    for (uint64_t p = off[i]; p < off[i + 1]; p++) best = std::min<int>(best, dist[p]);
    for (uint64_t p = off[i]; p < off[i + 1]; p++) {
        if (just && dist[p] != best) continue;
        const uint64_t key = hr.keys[ref[p]], rc = revcomp_key(key, k);
        bool placed = false;
        for (auto &e : kept)
            if (e.alive && e.key == rc) {
                if ((int)dist[p] < e.d) e.alive = false; else placed = true;
                break;
            }
        if (!placed) kept.push_back({key, ref[p], (int)dist[p], true});
    }
    struct Entry { std::vector<std::pair<int, int>> diffs; int64_t h; };
    std::vector<Entry> ents;
    const uint64_t skey = hs.keys[i];
    for (auto &e : kept) {
        if (!e.alive) continue;
        Entry en;
        en.h = e.h;
        for (int p = 0; p < k; p++) {
            const int a = (int)((skey >> (62 - 2 * p)) & 3), b = (int)((e.key >> (62 - 2 * p)) & 3);
            if (a != b) en.diffs.push_back({p, b});
        }
        ents.push_back(std::move(en));
    }
    std::stable_sort(ents.begin(), ents.end(), [](const Entry &x, const Entry &y) {
        if (x.diffs.size() != y.diffs.size()) return x.diffs.size() < y.diffs.size();
        return x.diffs < y.diffs;
    });
    bool first = true;
    for (auto &en : ents) {
        if (!first) s.push_back(' ');
        first = false;
        if (en.diffs.empty()) s.push_back('-');
        for (auto &d : en.diffs) { append_int(s, d.first); s.push_back("ACTG"[d.second]); }
        s.push_back('|');
        if (vals) append_values(s, *vals, (uint64_t)(*row_of)[en.h]);
        else append_values(s, hr, (uint64_t)en.h);
    }
}

int matcher(int argc, char **argv) {
    const Args a = parse_args(argc, argv, 2, "idozZntLPkK", "vj");
    const int max_diff = a.geti('n', 0);
    const bool just = a.has('j');
    if (max_diff < 0) throw std::runtime_error("-n must not be negative");
    StageClock clk(a.has('v'));
    if (!a.has('i') || !a.has('d') || !a.has('o')) throw std::runtime_error("-i, -d and -o are required");
    g_threads = std::max(1, a.geti('t', g_threads));
    Gpu g;
    const DatabaseSet ref_set(a.get('d')), search_set(a.get('i'));
    // -L start[-end]: only the search k-mers of keys start .. end (Matcher.java:159-171: end inclusive); a k-mer and
    // its match share their key, so the same slice of the reference is all that is needed.
    // Without -L the keys are walked in slices -- -P n (not in the reference), or as many as keep a slice's files
    // around 4 GB -- so that databases beyond one GPU database's 2^32 values, or the host's memory, join too; the
    // lines come out in the same order either way.
    int key_lo = 0, key_hi = std::max(ref_set.end_key, search_set.end_key);
    int slices = std::max<int>(std::max(1, a.geti('P', 1)), (int)((ref_set.bytes + search_set.bytes) >> 32) + 1);
    if (a.has('L')) {
        const std::string L = a.get('L');
        const size_t dash = L.find('-');
        key_lo = atoi(L.c_str());
        key_hi = (dash == std::string::npos ? key_lo : atoi(L.c_str() + dash + 1)) + 1;
        slices = 1;
    }
    slices = std::max(1, std::min(slices, key_hi - key_lo));
    FILE *f = fopen(a.get('o').c_str(), "wb");
    if (!f) throw std::runtime_error("cannot create " + a.get('o'));
    bool wrote = false;
    kmg_db *ref_whole = nullptr;
    for (int sl = 0; sl < slices; sl++) {
        const int lo = key_lo + (int)((int64_t)(key_hi - key_lo) * sl / slices);
        const int hi = key_lo + (int)((int64_t)(key_hi - key_lo) * (sl + 1) / slices);
        // (an inexact match can sit under any key close to the search k-mer's, KmerUtils.getCloseKeys: the whole
        // reference stays loaded then and only the search side is walked in slices)
        if (max_diff > 0 && !ref_whole) ref_whole = ref_set.load(g);
        kmg_db *ref = max_diff > 0 ? ref_whole : ref_set.load(g, lo, hi), *search = search_set.load(g, lo, hi);
        bool trie = false;
        // -k / -K default to the search file's lengths (Matcher.java:147-148).  One length L: both sides go through
        // KmerUtils.restrictedStream (KmerFile.allRestrictedKmers, Matcher.java:241; DB.java:134) -- kmg_db_restrict --
        // and take the exact join.  A search over several lengths is the trie (DB.java:83): kmg_match_trie.
        {
            const DbInfo s0 = db_info(search), r0 = db_info(ref);
            const int s_min = kmg_db_min_k(search), r_min = kmg_db_min_k(ref);
            const int want_min = a.geti('k', s_min), want_max = a.geti('K', s0.k);
            if (want_min < r_min || want_max > r0.k)                                                  // DB.java:78-81
                throw std::runtime_error("InconsistentDataException: Search kmers contain kmers of a length inconsistent with the database");
            if (want_min != want_max) {
                // a search over several lengths: the trie (DB.java:83,139-184) in closed form -- kmg_match_trie
                if (want_min != s_min || want_max != s0.k) {               // KmerFile.allRestrictedKmers (Matcher.java:241)
                    kmg_db *cut = nullptr;
                    g.check(kmg_db_restrict_range(g.ctx, search, want_min, want_max, &cut));
                    kmg_db_free(search);
                    search = cut;
                }
                trie = true;
            } else if (s_min != s0.k || s0.k != want_min) {
                kmg_db *cut = nullptr;
                g.check(kmg_db_restrict(g.ctx, search, want_min, &cut));
                kmg_db_free(search);
                search = cut;
            }
            if (!trie && (r_min != r0.k || r0.k != want_min)) {
                // (with differences too: the trie's data at depth L, shortest = L, is the sum over the k-mers sharing the
                // L-base prefix -- DESIGN 6b with m = L -- which is what the restricted database holds)
                kmg_db *cut = nullptr;
                g.check(kmg_db_restrict(g.ctx, ref, want_min, &cut));
                if (ref != ref_whole) kmg_db_free(ref);
                ref = cut;
            }
        }
        const DbInfo rd = db_info(ref), sd = db_info(search);
        if (!rd.rc)   // DB.java:57-64
            throw std::runtime_error("InconsistentDataException: Files contains different kmer parameters (min/max length, key length or reverse complement included");
        std::vector<int64_t> hit(sd.U + 1);
        uint64_t nh = 0;
        clk.lap("read + decode databases");
        // -n 1 | 2 [-j]: every reference k-mer within reach of every search k-mer (kmg_match_near), else the exact join
        std::vector<uint8_t> slens;                  // lengths of the search k-mers when there are several
        std::vector<uint64_t> pair_off;
        std::vector<int64_t> pair_ref;
        std::vector<uint8_t> pair_dist;
        std::vector<std::string> third;              // the lines' third field, where it is put together ahead of the formatting
        if (max_diff > 0 && trie) {
            // -n [-j] over several lengths (DB.java:139-184 over the trie): a search k-mer of T bases can meet the DISTINCT
            // T-base prefixes of the stored k-mers of at least T bases, and is handed for each what the trie holds at depth
            // T on that prefix's path.  Per length T: prefixes = the database restricted to T, their answers =
            // kmg_match_trie with the prefixes as the search side, neighbours = kmg_match_near against the prefixes.
            slens.resize(sd.U + 1);
            g.check(kmg_db_fetch_lens(search, slens.data()));
            HostDb hk;                               // (the search k-mers: only their keys are needed here)
            hk.d = sd;
            hk.keys.resize(sd.U + 1); hk.val_off.resize(sd.U + 1); hk.a.resize(sd.V + 1); hk.b.resize(sd.V + 1);
            g.check(kmg_db_fetch(search, hk.keys.data(), hk.val_off.data(), hk.a.data(), hk.b.data(), nullptr));
            third.assign(sd.U, std::string());
            const int shortest = kmg_db_min_k(search);
            for (int T = shortest; T <= sd.k; T++) {
                std::vector<uint64_t> sel;
                for (uint64_t i = 0; i < sd.U; i++) if (slens[i] == T) sel.push_back(i);
                if (sel.empty()) continue;
                HostDb sub;
                sub.keys.resize(sel.size());
                std::vector<uint64_t> voff(sel.size() + 1);
                std::vector<int32_t> zero(sel.size() + 1, 0);
                for (size_t j = 0; j < sel.size(); j++) { sub.keys[j] = hk.keys[sel[j]]; voff[j] = j; }
                voff[sel.size()] = sel.size();
                kmg_db *subdb = nullptr, *pre = ref, *rowsdb = nullptr;
                g.check(kmg_db_load(g.ctx, T, sd.key_len, 0, KMG_PAYLOAD_READPOS, sub.keys.data(), sel.size(), voff.data(), zero.data(),
                                    zero.data(), &subdb));
                if (kmg_db_min_k(ref) != rd.k || rd.k != T) g.check(kmg_db_restrict(g.ctx, ref, T, &pre));
                const DbInfo pd = db_info(pre);
                std::vector<int64_t> row_of(pd.U + 1);
                uint64_t nrow = 0, np = 0;
                g.check(kmg_match_trie(g.ctx, pre, ref, shortest, 0, &rowsdb, row_of.data(), &nrow));
                const HostDb hpre = fetch(g, pre), hrows = fetch(g, rowsdb);
                std::vector<uint64_t> p_off(sel.size() + 1);
                g.check(kmg_match_near(g.ctx, subdb, pre, max_diff, p_off.data(), nullptr, nullptr, 0, &np));
                std::vector<int64_t> p_ref(np + 1);
                std::vector<uint8_t> p_dist(np + 1);
                if (np) g.check(kmg_match_near(g.ctx, subdb, pre, max_diff, nullptr, p_ref.data(), p_dist.data(), np, &np));
                parallel_for((sel.size() + 4095) / 4096, [&](size_t c) {
                    for (size_t j = c * 4096; j < std::min(sel.size(), (c + 1) * 4096); j++)
                        if (p_off[j + 1] > p_off[j]) append_near(third[sel[j]], sub, hpre, j, T, p_off, p_ref, p_dist, just, &hrows, &row_of);
                });
                kmg_db_free(subdb);
                kmg_db_free(rowsdb);
                if (pre != ref) kmg_db_free(pre);
            }
            for (uint64_t i = 0; i < sd.U; i++) { hit[i] = third[i].empty() ? -1 : 0; nh += hit[i] >= 0; }
        } else if (max_diff > 0) {
            uint64_t np = 0;
            pair_off.resize(sd.U + 1);
            g.check(kmg_match_near(g.ctx, search, ref, max_diff, pair_off.data(), nullptr, nullptr, 0, &np));
            pair_ref.resize(np + 1);
            pair_dist.resize(np + 1);
            if (np) g.check(kmg_match_near(g.ctx, search, ref, max_diff, nullptr, pair_ref.data(), pair_dist.data(), np, &np));
            for (uint64_t i = 0; i < sd.U; i++) { hit[i] = pair_off[i + 1] > pair_off[i] ? 0 : -1; nh += hit[i] >= 0; }
        } else if (trie) {
            // the answers come as rows of a database of their own (counts already summed per taxon)
            kmg_db *rows = nullptr;
            g.check(kmg_match_trie(g.ctx, search, ref, kmg_db_min_k(search), 0, &rows, hit.data(), &nh));
            kmg_db_free(ref);
            ref = rows;
            slens.resize(sd.U + 1);
            g.check(kmg_db_fetch_lens(search, slens.data()));
        } else g.check(kmg_match(g.ctx, search, ref, hit.data(), &nh));
        const HostDb hs = fetch(g, search), hr = fetch(g, ref);
        if (ref != ref_whole) kmg_db_free(ref);
        kmg_db_free(search);
        clk.lap("match (GPU) + fetch");
        // one line per matched search k-mer, ascending (Matcher.java:247-249, ResultsDataType.java:15-35):
        //   KMER \t values of the search k-mer \t -|values of the reference k-mer       ("-" = empty KmerDiff)
        std::vector<uint64_t> rows;
        rows.reserve(nh);
        for (uint64_t i = 0; i < sd.U; i++) if (hit[i] >= 0) rows.push_back(i);
        const size_t nchunk = (rows.size() + 65535) / 65536;
        std::vector<std::string> text(nchunk);
        parallel_for(nchunk, [&](size_t c) {
            std::string &s = text[c];
            const size_t lo_r = c * 65536, hi_r = std::min(rows.size(), lo_r + 65536);
            s.reserve((hi_r - lo_r) * (size_t)(sd.k + 48));
            for (size_t r = lo_r; r < hi_r; r++) {
                const uint64_t i = rows[r], key = hs.keys[i];
                const int len = trie ? (int)slens[i] : sd.k;
                for (int p = 0; p < len; p++) s.push_back("ACTG"[(key >> (62 - 2 * p)) & 3]);   // Sequence.java:117-134
                s.push_back('\t');
                append_values(s, hs, i);
                if (max_diff == 0) {
                    s += "\t-|";
                    append_values(s, hr, (uint64_t)hit[i]);
                } else if (trie) {
                    s.push_back('\t');
                    s += third[i];
                } else {
                    s.push_back('\t');
                    append_near(s, hs, hr, i, sd.k, pair_off, pair_ref, pair_dist, just);
                }
                s.push_back('\n');
            }
        });
        clk.lap("format match lines");
        if (a.has('Z')) {
            for (auto &s : text) fwrite(s.data(), 1, s.size(), f);
        } else {
            // gzip at the default level (Matcher.java:138-139).  One member per 64 K lines, deflated on the pool:
            // a multi-member file is what every gzip reader (java.util.zip.GZIPInputStream included) accepts, and
            // single-threaded deflate of the match text would otherwise dominate the whole program.
            std::vector<std::string> zipped(nchunk);
            parallel_for(nchunk, [&](size_t c) {
                zipped[c] = gzip_member((const uint8_t *)text[c].data(), text[c].size(), 6);
                std::string().swap(text[c]);
            });
            for (auto &s : zipped) { fwrite(s.data(), 1, s.size(), f); wrote = wrote || !s.empty(); }
        }
        clk.lap("deflate + write");
    }
    if (ref_whole) kmg_db_free(ref_whole);
    if (!a.has('Z') && !wrote) { const std::string e = gzip_member(nullptr, 0, 6); fwrite(e.data(), 1, e.size(), f); }
    fclose(f);
    return 0;
}

int taxa_counts(int argc, char **argv) {
    const Args a = parse_args(argc, argv, 2, "dotP", "");
    if (!a.has('d') || !a.has('o')) throw std::runtime_error("-d and -o are required");
    g_threads = std::max(1, a.geti('t', g_threads));
    Gpu g;
    // the keys in slices, as in Matcher (-P n, or by file size): a k-mer lives in one slice, so the counts add
    const DatabaseSet set(a.get('d'));
    const int slices = std::max(1, std::min(std::max<int>(std::max(1, a.geti('P', 1)), (int)(set.bytes >> 32) + 1), set.end_key));
    std::map<int32_t, std::pair<int64_t, int64_t>> total;
    for (int sl = 0; sl < slices; sl++) {
        const int lo = (int)((int64_t)set.end_key * sl / slices), hi = (int)((int64_t)set.end_key * (sl + 1) / slices);
        kmg_db *ref = set.load(g, lo, hi);
        const DbInfo d = db_info(ref);
        uint64_t n = std::max<uint64_t>(d.V, 1);
        std::vector<int32_t> t(n);
        std::vector<int64_t> m(n), q(n);
        g.check(kmg_taxacounts(g.ctx, ref, t.data(), m.data(), q.data(), &n));
        kmg_db_free(ref);
        for (uint64_t i = 0; i < n; i++) { auto &e = total[t[i]]; e.first += m[i]; e.second += q[i]; }
    }
    FILE *f = fopen(a.get('o').c_str(), "w");
    if (!f) throw std::runtime_error("cannot create " + a.get('o'));
    for (auto &kv : total) fprintf(f, "%d\t%lld\t%lld\n", kv.first, (long long)kv.second.first, (long long)kv.second.second);   // TaxaCounts.java:142-150
    fclose(f);
    return 0;
}

// ---------------------------------------------------------------------------------------------- per-read tools
// new IndexedOutputFile<>(f, new IntCompressor(), true, 5): human-readable flavour, zip level 5, one block per key
// ascending, no meta block (CollectByRead.java:66-67,116-119; ReadClassifier.java:52-53,157-160)
void write_text_blocks(const std::string &path, const std::vector<std::pair<int, std::string>> &blocks) {
    std::vector<std::string> blobs(blocks.size());
    parallel_for(blocks.size(), [&](size_t i) {
        blobs[i] = gzip_member((const uint8_t *)blocks[i].second.data(), blocks[i].second.size(), 5);
    });
    IndexedOutputFile out(path, 5, true);
    for (size_t i = 0; i < blocks.size(); i++) out.add_block(blocks[i].first, blobs[i]);
    out.close();
}

// the two databases a list of (k-mer | line, occurrences, taxa) rows stands for
struct JoinRows {
    std::vector<uint64_t> keys, s_off{0}, r_off{0};
    std::vector<int32_t> s_read, s_pos, r_tax, r_cnt;
    void load(Gpu &g, int k, int key_len, kmg_db **search, kmg_db **ref) {
        s_read.push_back(0); s_pos.push_back(0); r_tax.push_back(0); r_cnt.push_back(0);     // (never an empty array)
        g.check(kmg_db_load(g.ctx, k, key_len, 0, KMG_PAYLOAD_READPOS, keys.data(), keys.size(), s_off.data(), s_read.data(),
                            s_pos.data(), search));
        g.check(kmg_db_load(g.ctx, k, key_len, 1, KMG_PAYLOAD_TAXON, keys.data(), keys.size(), r_off.data(), r_tax.data(),
                            r_cnt.data(), ref));
    }
};

// TxC|TxC... up to `end` (CountDataType.fromString, separators "x" and "|")
void parse_counts(const char *q, const char *end, JoinRows &j) {
    while (q < end) {
        char *e;
        const long t = strtol(q, &e, 10);
        if (e >= end || *e != 'x') throw std::runtime_error("bad taxon count list");
        const long c = strtol(e + 1, &e, 10);
        j.r_tax.push_back((int32_t)t);
        j.r_cnt.push_back((int32_t)std::min<long>(c, 0x7FFFFFFF));
        q = (e < end && *e == '|') ? e + 1 : end;
    }
    j.r_off.push_back(j.r_tax.size());
}

// The third field of a match line with inexact entries, `DIFF|TxC|TxC DIFF|TxC ...`, as CollectByRead.processKmer
// (CollectByRead.java:165-173) files it: the taxa of all entries with the same number of differences added up
// (KmerDiff.dist = number of position + base pairs, "-" = none), printed `dist:TxC|TxC` per distance, ascending, joined
// by " " (jar MapCompressor text form), counts capped at 255 on the way through the temp file (CountDataType.java:40).
std::string dist_taxa_text(const char *q, const char *end) {
    std::map<int, std::map<int32_t, int64_t>> by_dist;
    while (q < end) {
        const char *sp = (const char *)memchr(q, ' ', end - q);
        const char *ee = sp ? sp : end;
        const char *bar = (const char *)memchr(q, '|', ee - q);
        if (!bar) throw std::runtime_error("bad match entry");
        int dist = 0;
        for (const char *c = q; c < bar; c++) dist += (*c >= 'A' && *c <= 'Z');
        auto &m = by_dist[dist];
        for (const char *t = bar + 1; t < ee;) {
            char *e;
            const long taxon = strtol(t, &e, 10);
            if (e >= ee || *e != 'x') throw std::runtime_error("bad taxon count list");
            const long cnt = strtol(e + 1, &e, 10);
            m[(int32_t)taxon] += cnt;
            t = (e < ee && *e == '|') ? e + 1 : ee;
        }
        q = sp ? sp + 1 : end;
    }
    std::string out;
    for (auto &d : by_dist) {
        if (!out.empty()) out += ' ';
        append_int(out, d.first);
        out += ':';
        bool first = true;
        for (auto &tc : d.second) {
            if (!first) out += '|';
            first = false;
            append_int(out, tc.first);
            out += 'x';
            append_int(out, (int32_t)std::min<int64_t>(tc.second, 255));
        }
    }
    return out;
}

// Utils.CollectByRead -i <Matcher output> -o <file> (CollectByRead.java:28-90): the match lines are loaded back as
// a read database and a reference database over the same k-mers (every k-mer hits itself) and transposed on the GPU.
int collect_by_read(int argc, char **argv) {
    const Args a = parse_args(argc, argv, 2, "ioft", "v");
    if (!a.has('i') || !a.has('o')) throw std::runtime_error("-i and -o are required");
    g_threads = std::max(1, a.geti('t', g_threads));
    StageClock clk(a.has('v'));
    const std::string text = read_input(a.get('i'));
    clk.lap("read + inflate match text");
    // the lines are parsed in slices cut at line ends, side by side, and the slices' arrays laid end to end
    const size_t nslice = std::max<size_t>(1, std::min<size_t>((size_t)g_threads * 4, text.size() / (1 << 20)));
    std::vector<size_t> cut(nslice + 1, text.size());
    cut[0] = 0;
    for (size_t c = 1; c < nslice; c++) {
        const size_t at = text.find('\n', std::max(cut[c - 1], text.size() / nslice * c));
        cut[c] = at == std::string::npos ? text.size() : at + 1;
    }
    std::vector<JoinRows> part(nslice);
    std::vector<int> part_k(nslice, -1);
    std::vector<std::string> unsupported(nslice);
    std::vector<std::vector<std::pair<uint64_t, std::string>>> special(nslice);
    parallel_for(nslice, [&](size_t c) {
        JoinRows &j = part[c];
        int &k = part_k[c];
        size_t p0 = cut[c];
        const size_t stop = cut[c + 1];
        while (p0 < stop) {
            size_t e = text.find('\n', p0);
            if (e == std::string::npos) e = text.size();
            if (e > p0) {
                const char *line = text.data() + p0, *lend = text.data() + e;
                const char *t1 = (const char *)memchr(line, '\t', lend - line);
                const char *t2 = t1 ? (const char *)memchr(t1 + 1, '\t', lend - t1 - 1) : nullptr;
                if (!t2) throw std::runtime_error("bad match line");
                if (k == -1) k = (int)(t1 - line);
                // k-mers of several lengths (Matcher's default on default databases) or beyond 32 bases: the k-mer only
                // names the row here, so the line number stands in for it (k = -2; as read_classifier does throughout)
                if ((int)(t1 - line) != k || k < 1 || k > 32) k = -2;
                uint64_t v = 0;
                for (const char *ch = line; k > 0 && ch < t1; ch++) {
                    int b;
                    switch (*ch) { case 'A': b = 0; break; case 'C': b = 1; break; case 'T': b = 2; break; case 'G': b = 3; break;
                                   default: throw std::runtime_error("bad k-mer in match line"); }
                    v = (v << 2) | (uint64_t)b;
                }
                j.keys.push_back(k < 0 || k == 32 ? v : v << (64 - 2 * k));
                for (const char *q = t1 + 1; q < t2;) {                       // r:p|r:p
                    char *en;
                    const long r = strtol(q, &en, 10);
                    if (*en != ':') throw std::runtime_error("bad read position list");
                    const long ps = strtol(en + 1, &en, 10);
                    j.s_read.push_back((int32_t)r);
                    j.s_pos.push_back((int32_t)ps);
                    q = (*en == '|') ? en + 1 : t2;
                }
                j.s_off.push_back(j.s_read.size());
                // one entry with diff "-" (an exact match): its taxa go into the arrays; anything else (Matcher -n > 0)
                // is turned into the record's text right here and the line keeps an empty taxon list as its mark
                if (t2 + 2 < lend && t2[1] == '-' && t2[2] == '|' && !memchr(t2 + 1, ' ', lend - t2 - 1)) parse_counts(t2 + 3, lend, j);
                else {
                    j.r_off.push_back(j.r_tax.size());
                    special[c].push_back({j.keys.size() - 1, dist_taxa_text(t2 + 1, lend)});
                }
            }
            p0 = e + 1;
        }
    });
    JoinRows j;
    std::unordered_map<uint64_t, std::string> special_text;      // line -> `dist:TxC|TxC dist:...` where it is not one exact match
    int k = -1;
    for (size_t c = 0; c < nslice; c++) {
        if (!unsupported[c].empty()) throw Unsupported(unsupported[c]);
        if (part_k[c] == -1) continue;
        if (k == -1) k = part_k[c];
        if (part_k[c] != k) k = -2;
    }
    {
        std::vector<uint64_t> kb(nslice + 1, 0), sb(nslice + 1, 0), rb(nslice + 1, 0);
        for (size_t c = 0; c < nslice; c++) {
            kb[c + 1] = kb[c] + part[c].keys.size();
            sb[c + 1] = sb[c] + part[c].s_read.size();
            rb[c + 1] = rb[c] + part[c].r_tax.size();
        }
        j.keys.resize(kb[nslice]); j.s_off.resize(kb[nslice] + 1); j.r_off.resize(kb[nslice] + 1);
        j.s_read.resize(sb[nslice]); j.s_pos.resize(sb[nslice]); j.r_tax.resize(rb[nslice]); j.r_cnt.resize(rb[nslice]);
        parallel_for(nslice, [&](size_t c) {
            JoinRows &q = part[c];
            std::copy(q.keys.begin(), q.keys.end(), j.keys.begin() + kb[c]);
            for (size_t i = 0; i < q.keys.size(); i++) {
                j.s_off[kb[c] + i] = sb[c] + q.s_off[i];
                j.r_off[kb[c] + i] = rb[c] + q.r_off[i];
            }
            std::copy(q.s_read.begin(), q.s_read.end(), j.s_read.begin() + sb[c]);
            std::copy(q.s_pos.begin(), q.s_pos.end(), j.s_pos.begin() + sb[c]);
            std::copy(q.r_tax.begin(), q.r_tax.end(), j.r_tax.begin() + rb[c]);
            std::copy(q.r_cnt.begin(), q.r_cnt.end(), j.r_cnt.begin() + rb[c]);
            q = JoinRows();
        });
        j.s_off[kb[nslice]] = sb[nslice];
        j.r_off[kb[nslice]] = rb[nslice];
        if (k == -2) {                       // rows named by their line numbers, as 32-mers
            for (size_t i = 0; i < j.keys.size(); i++) j.keys[i] = i;
            k = 32;
        }
        for (size_t c = 0; c < nslice; c++)
            for (auto &sp : special[c]) special_text.emplace(kb[c] + sp.first, std::move(sp.second));
    }
    clk.lap("parse match lines");
    std::vector<std::pair<int, std::string>> blocks;
    if (!j.keys.empty()) {
        Gpu g;
        kmg_db *search = nullptr, *ref = nullptr;
        j.load(g, k, std::min(k, 6), &search, &ref);
        const uint64_t cap = std::max<uint64_t>(j.s_read.size(), 1);
        std::vector<int32_t> read(cap), pos(cap);
        std::vector<int64_t> idx(cap);
        uint64_t n = 0;
        g.check(kmg_collect_by_read(g.ctx, search, ref, &n, read.data(), pos.data(), idx.data()));
        kmg_db_free(search);
        kmg_db_free(ref);
        clk.lap("load + transpose (GPU)");
        // block bounds: (read / 1000) * 1000; of two records with one (read, pos) the first is kept (TreeSet.add)
        std::vector<uint64_t> starts;
        for (uint64_t i = 0; i < n; i++)
            if (i == 0 || read[i] / 1000 != read[i - 1] / 1000) starts.push_back(i);
        starts.push_back(n);
        blocks.resize(starts.size() - 1);
        parallel_for(blocks.size(), [&](size_t b) {
            std::string out;
            out.reserve((starts[b + 1] - starts[b]) * 32);
            for (uint64_t i = starts[b]; i < starts[b + 1]; i++) {
                // the hit k-mers' taxa lie anywhere in the reference arrays: fetch ahead
                if (i + 16 < starts[b + 1]) __builtin_prefetch(&j.r_off[idx[i + 16]]);
                if (i + 8 < starts[b + 1]) { __builtin_prefetch(&j.r_tax[j.r_off[idx[i + 8]]]); __builtin_prefetch(&j.r_cnt[j.r_off[idx[i + 8]]]); }
                if (i > starts[b] && read[i] == read[i - 1] && pos[i] == pos[i - 1]) continue;
                if (!out.empty()) out += '\n';
                append_int(out, read[i]);
                out += ':';
                append_int(out, pos[i]);
                if (j.r_off[idx[i]] == j.r_off[idx[i] + 1]) {          // (an exact match always names a taxon)
                    out += '\t';
                    out += special_text.at((uint64_t)idx[i]);
                    continue;
                }
                out += "\t0:";
                for (uint64_t t = j.r_off[idx[i]]; t < j.r_off[idx[i] + 1]; t++) {
                    if (t > j.r_off[idx[i]]) out += '|';
                    append_int(out, j.r_tax[t]);
                    out += 'x';
                    append_int(out, std::min(j.r_cnt[t], 255));
                }
            }
            blocks[b] = {(read[starts[b]] / 1000) * 1000, std::move(out)};
        });
        clk.lap("format lines");
    }
    write_text_blocks(a.get('o'), blocks);
    clk.lap("deflate + write");
    return 0;
}

// Utils.ReadClassifier -i <CollectByRead output> -o <file> (ReadClassifier.java:28-78): every line becomes one
// pseudo k-mer (its line number as a 32-mer) of a read database holding the line's (read, pos) and of a reference
// database holding its dist-0 taxa; kmg_classify_reads then does processRead for all reads at once.
int read_classifier(int argc, char **argv) {
    const Args a = parse_args(argc, argv, 2, "iot", "v");
    if (!a.has('i') || !a.has('o')) throw std::runtime_error("-i and -o are required");
    g_threads = std::max(1, a.geti('t', g_threads));
    StageClock clk(a.has('v'));
    IndexedInputFile in(a.get('i'));
    // every block is inflated and parsed on its own, side by side; the blocks' rows are then laid end to end
    std::vector<int> block_key;
    for (auto &kv : in.index) block_key.push_back(kv.first);
    const size_t nb = block_key.size();
    std::vector<JoinRows> part(nb);
    std::vector<std::pair<int, std::vector<int32_t>>> block_reads(nb);
    std::vector<int32_t> part_max(nb, -1);
    parallel_for(nb, [&](size_t bi) {
        const std::string text = in.read_key(block_key[bi]);
        JoinRows &j = part[bi];
        block_reads[bi].first = block_key[bi];
        auto &reads = block_reads[bi].second;
        int32_t &max_read = part_max[bi];
        size_t p0 = 0;
        while (p0 < text.size()) {
            size_t e = text.find('\n', p0);
            if (e == std::string::npos) e = text.size();
            if (e > p0) {
                const char *line = text.data() + p0, *lend = text.data() + e;
                char *en;
                const long r = strtol(line, &en, 10);
                if (*en != ':') throw std::runtime_error("bad CollectByRead line");
                const long ps = strtol(en + 1, &en, 10);
                if (*en != '\t') throw std::runtime_error("bad CollectByRead line");
                if (reads.empty() || reads.back() != (int32_t)r) reads.push_back((int32_t)r);
                for (const char *q = en + 1; q < lend;) {                  // dist:TxC|TxC entries joined by " "
                    const char *sp = (const char *)memchr(q, ' ', lend - q);
                    const char *ee = sp ? sp : lend;
                    char *c;
                    const long dist = strtol(q, &c, 10);
                    if (*c != ':') throw std::runtime_error("bad CollectByRead line");
                    if (dist == 0) {                                        // ReadClassifier.java:109,125
                        j.s_read.push_back((int32_t)r);      // (keys and offsets are the line numbers: filled in below)
                        j.s_pos.push_back((int32_t)ps);
                        parse_counts(c + 1, ee, j);
                        max_read = std::max(max_read, (int32_t)r);
                    }
                    q = sp ? sp + 1 : lend;
                }
            }
            p0 = e + 1;
        }
    });
    JoinRows j;
    int32_t max_read = -1;
    {
        std::vector<uint64_t> kb(nb + 1, 0), rb(nb + 1, 0);
        for (size_t c = 0; c < nb; c++) {
            kb[c + 1] = kb[c] + part[c].s_read.size();
            rb[c + 1] = rb[c] + part[c].r_tax.size();
            max_read = std::max(max_read, part_max[c]);
        }
        j.keys.resize(kb[nb]); j.s_off.resize(kb[nb] + 1); j.r_off.resize(kb[nb] + 1);
        j.s_read.resize(kb[nb]); j.s_pos.resize(kb[nb]); j.r_tax.resize(rb[nb]); j.r_cnt.resize(rb[nb]);
        parallel_for(nb, [&](size_t c) {
            JoinRows &q = part[c];
            for (size_t i = 0; i < q.s_read.size(); i++) {
                j.keys[kb[c] + i] = kb[c] + i;                   // one pseudo k-mer per line: its number
                j.s_off[kb[c] + i] = kb[c] + i;
                j.r_off[kb[c] + i] = rb[c] + q.r_off[i];
            }
            std::copy(q.s_read.begin(), q.s_read.end(), j.s_read.begin() + kb[c]);
            std::copy(q.s_pos.begin(), q.s_pos.end(), j.s_pos.begin() + kb[c]);
            std::copy(q.r_tax.begin(), q.r_tax.end(), j.r_tax.begin() + rb[c]);
            std::copy(q.r_cnt.begin(), q.r_cnt.end(), j.r_cnt.begin() + rb[c]);
            q = JoinRows();
        });
        j.s_off[kb[nb]] = kb[nb];
        j.r_off[kb[nb]] = rb[nb];
    }
    clk.lap("read + inflate + parse");
    std::vector<int32_t> call((size_t)std::max<int32_t>(max_read + 1, 1), -1);
    if (!j.keys.empty()) {
        Gpu g;
        kmg_db *search = nullptr, *ref = nullptr;
        j.load(g, 32, 6, &search, &ref);
        g.check(kmg_classify_reads(g.ctx, search, ref, (uint64_t)max_read + 1, call.data()));
        kmg_db_free(search);
        kmg_db_free(ref);
    }
    std::vector<std::pair<int, std::string>> blocks;
    for (auto &br : block_reads) {
        std::string out;
        char buf[48];
        for (int32_t r : br.second)
            if (r >= 0 && r <= max_read && call[r] >= 0) out.append(buf, snprintf(buf, sizeof buf, "%d\t%d\n", r, call[r]));
        blocks.push_back({br.first, std::move(out)});
    }
    clk.lap("classify (GPU) + format");
    write_text_blocks(a.get('o'), blocks);
    clk.lap("deflate + write");
    return 0;
}

// Pipeline.ReadCalls -q reads.fq (-d database | -a reference.fa) -o calls [-K maxK -k minK -l key_length -r read_map.gz -C seqs -B bytes]
// The file the chain MakeDatabase -q | Matcher | CollectByRead | ReadClassifier ends with, byte for byte, without the
// three files in between: both databases stay on the GPU, the join feeds kmg_classify_reads_hits and only the
// `read \t call` lines leave.  No counterpart program in the reference.
int read_calls(int argc, char **argv) {
    const Args a = parse_args(argc, argv, 2, "qdaoKklrtCB", "v");
    if (!a.has('q') || !a.has('o') || a.has('d') == a.has('a'))
        throw std::runtime_error("-q, -o and exactly one of -d (database) / -a (FASTA reference) are required");
    g_threads = std::max(1, a.geti('t', g_threads));
    StageClock clk(a.has('v'));
    // The reads come in pieces of the file (-B bytes, default 16 MiB): a reader thread inflates and splits piece i + 1
    // while the GPU builds, joins and classifies piece i, and neither the text nor the sequences of the whole input are
    // ever in memory.  A call depends on its own read only.  Reads number on across pieces as ReadIDMapping would: the
    // sequences that share a piece's LAST read index may have more of their kind behind the cut, so they are carried
    // over and lead the next build -- they hand out k-mers, so the numbering there starts as it would have gone on.
    // -C n: at most n sequences of a piece per build.
    const uint64_t piece_bytes = a.has('B') ? std::max<uint64_t>(1, size_from_human(a.get('B'))) : (16ull << 20);
    // (the reader starts before the GPU context and the database are up -- FASTQ splitting needs neither)
    const uint64_t chunk_seqs = a.has('C') ? (uint64_t)std::max(1, a.geti('C', 1)) : 0;
    PieceReader reader(a.get('q'), 1, 0, piece_bytes);
    Gpu g;
    clk.lap("GPU context");
    int k = a.geti('K', 32), key_len = a.geti('l', 6);
    int min_k = a.geti('k', k);
    if (min_k > k) throw std::runtime_error("-k (shortest k-mer) above -K (longest)");
    kmg_db *ref = nullptr;
    if (a.has('d')) {
        ref = open_database_set(g, a.get('d'));
        const DbInfo rd = db_info(ref);
        k = rd.k; key_len = rd.key_len;
        min_k = kmg_db_min_k(ref);       // the reads are built with the database's lengths, as MakeDatabase -q -K -k would
        clk.lap("read + decode database");
    } else ref = build_from_text(g, a.get('a'), false, k, key_len, false, -1, -1, "", nullptr, nullptr, clk, 1, nullptr, 0, 1, min_k);
    // k-mers of several lengths (the reference's defaults, -K 32 -k 24): Matcher's search is then the trie's
    // (DB.java:83,139-184 -> kmg_match_trie) and the calls come from its answers (kmg_classify_reads_rows)
    const bool several = min_k != k;
    const int count_cap = a.has('d') ? 0 : 255;
    if (!db_info(ref).rc)   // DB.java:57-64
        throw std::runtime_error("InconsistentDataException: Files contains different kmer parameters (min/max length, key length or reverse complement included");
    gzFile zmap = nullptr;
    if (a.has('r')) {
        zmap = gzopen(a.get('r').c_str(), "wb");
        if (!zmap) throw std::runtime_error("cannot create " + a.get('r'));
    }
    std::vector<int32_t> call;
    std::vector<uint8_t> seen;
    // sequences carried over from the build before (those of its last read index)
    std::vector<uint8_t> left_bases;
    std::vector<uint64_t> left_off{0};
    std::vector<int64_t> left_tok;
    std::vector<std::string> left_names;
    auto run_build = [&](const kmh_batch *b, uint64_t s0, uint64_t s1, bool last_of_input) {
        const uint64_t *off = kmh_seq_off(b);
        const int64_t *tok = kmh_seq_tok(b);
        const uint64_t *noff = kmh_name_off(b);
        const char *names = kmh_names(b);
        const uint64_t nl = left_tok.size(), np = s1 - s0;
        if (nl + np == 0) return;
        std::vector<int32_t> ridx(nl + np, -1);
        g.check(kmg_build_begin(g.ctx, k, key_len, 0, KMG_PAYLOAD_READPOS));
        if (several) g.check(kmg_build_set_min_k(g.ctx, min_k));
        if (nl) g.check(kmg_build_add(g.ctx, left_bases.data(), left_bases.size(), left_off.data(), nullptr, left_tok.data(), nl, ridx.data()));
        std::vector<uint64_t> loff(np + 1);
        if (np) {
            for (uint64_t i = 0; i <= np; i++) loff[i] = off[s0 + i] - off[s0];
            g.check(kmg_build_add(g.ctx, kmh_bases(b) + off[s0], loff.back(), loff.data(), nullptr, tok + s0, np, ridx.data() + nl));
        }
        kmg_db *reads = nullptr;
        g.check(kmg_build_finish(g.ctx, &reads, nullptr));
        int32_t top = -1;
        for (int32_t r : ridx) top = std::max(top, r);
        const uint64_t n_local = (uint64_t)(top + 1);
        uint64_t keep = n_local, next = nl + np;
        if (!last_of_input && n_local > 0) {
            keep = n_local - 1;
            next = 0;
            while (ridx[next] != top) next++;
        }
        if (keep) {
            std::vector<int32_t> c(n_local, -1);
            std::vector<uint8_t> h(n_local, 0);
            if (several) {
                kmg_db *rows = nullptr;
                std::vector<int64_t> hit_row(db_info(reads).U + 1);
                g.check(kmg_match_trie(g.ctx, reads, ref, min_k, count_cap, &rows, hit_row.data(), nullptr));
                const int rc2 = kmg_classify_reads_rows(g.ctx, reads, rows, hit_row.data(), n_local, c.data(), h.data());
                kmg_db_free(rows);
                g.check(rc2);
            } else
                g.check(kmg_classify_reads_hits(g.ctx, reads, ref, n_local, c.data(), h.data()));
            call.insert(call.end(), c.begin(), c.begin() + keep);
            seen.insert(seen.end(), h.begin(), h.begin() + keep);
        }
        kmg_db_free(reads);
        if (zmap) {
            int32_t last = -1;
            for (uint64_t s = 0; s < nl + np; s++) {
                const int32_t r = ridx[s];
                if (r >= 0 && r != last && (uint64_t)r < keep) {
                    if (s < nl) gzwrite(zmap, left_names[s].data(), (unsigned)left_names[s].size());
                    else gzwrite(zmap, names + noff[s0 + s - nl], (unsigned)(noff[s0 + s - nl + 1] - noff[s0 + s - nl]));
                    gzwrite(zmap, "\n", 1);
                }
                if (r >= 0) last = r;
            }
        }
        // what is carried over: the combined sequences [next, nl + np)
        std::vector<uint8_t> nb;
        std::vector<uint64_t> no{0};
        std::vector<int64_t> nt;
        std::vector<std::string> nn;
        for (uint64_t s = next; s < nl + np; s++) {
            if (s < nl) {
                nb.insert(nb.end(), left_bases.begin() + left_off[s], left_bases.begin() + left_off[s + 1]);
                nt.push_back(left_tok[s]);
                nn.push_back(left_names[s]);
            } else {
                const uint64_t q = s0 + s - nl;
                nb.insert(nb.end(), kmh_bases(b) + off[q], kmh_bases(b) + off[q + 1]);
                nt.push_back(tok[q]);
                nn.emplace_back(names + noff[q], noff[q + 1] - noff[q]);
            }
            no.push_back(nb.size());
        }
        left_bases.swap(nb); left_off.swap(no); left_tok.swap(nt); left_names.swap(nn);
    };
    for (bool done = false; !done;) {
        Piece p = reader.next();
        const kmh_batch *b = p.batch.get();
        if (kmh_high_bytes(b)) throw Unsupported("sequence lines hold bytes >= 128: not supported on the GPU path");
        if (kmh_empty_header(b)) throw Unsupported("a FASTQ record with an empty header: not supported on the GPU path");
        if (p.last && kmh_eof_in_kmer(b)) throw Unsupported("input ends inside a sequence line (no final newline): not supported on the GPU path for FASTQ input");
        const uint64_t nseq = kmh_nseq(b);
        const uint64_t step = chunk_seqs ? chunk_seqs : std::max<uint64_t>(nseq, 1);
        for (uint64_t s0 = 0; s0 < nseq || (s0 == 0 && p.last); s0 += step) {
            const uint64_t s1 = std::min(nseq, s0 + step);
            run_build(b, s0, s1, p.last && s1 == nseq);
            if (nseq == 0) break;
        }
        done = p.last;
    }
    if (zmap) gzclose(zmap);
    kmg_db_free(ref);
    const uint64_t n_reads = call.size();
    clk.lap("read + split (reader thread) alongside build + join + classify (GPU)");
    // one block per (read / 1000) * 1000 that holds a read with a matched k-mer (CollectByRead.java:105-124 writes no
    // other); inside, the reads that have a call (ReadClassifier.java:140-160)
    std::vector<std::pair<int, std::string>> blocks;
    for (uint64_t base = 0; base < n_reads; base += 1000) {
        const uint64_t end = std::min(n_reads, base + 1000);
        bool any = false;
        std::string out;
        char buf[48];
        for (uint64_t r = base; r < end; r++) {
            any |= seen[r] != 0;
            if (call[r] >= 0) out.append(buf, snprintf(buf, sizeof buf, "%llu\t%d\n", (unsigned long long)r, call[r]));
        }
        if (any) blocks.push_back({(int)base, std::move(out)});
    }
    write_text_blocks(a.get('o'), blocks);
    clk.lap("write calls");
    return 0;
}

}  // namespace

int main(int argc, char **argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: kmg Database.MakeDatabase|Database.Matcher|Utils.TaxaCounts|Utils.CollectByRead|Utils.ReadClassifier|Pipeline.ReadCalls <options>\n");
        return 1;
    }
    std::string prog = argv[1];
    const size_t dot = prog.rfind('.');
    if (dot != std::string::npos) prog = prog.substr(dot + 1);
    try {
        if (prog == "MakeDatabase") return make_database(argc, argv);
        if (prog == "Matcher") return matcher(argc, argv);
        if (prog == "TaxaCounts") return taxa_counts(argc, argv);
        if (prog == "CollectByRead") return collect_by_read(argc, argv);
        if (prog == "ReadClassifier") return read_classifier(argc, argv);
        if (prog == "ReadCalls") return read_calls(argc, argv);
        fprintf(stderr, "unknown program %s\n", argv[1]);
        return 1;
    } catch (const Unsupported &e) {
        fprintf(stderr, "kmg: not on the GPU path: %s\n", e.what());
        return 3;
    } catch (const std::exception &e) {
        fprintf(stderr, "kmg: %s\n", e.what());
        return 1;
    }
}
