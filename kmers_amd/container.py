"""The reference's on-disk container (jar IndexedFiles.*), host-side I/O.

File = fixed header | data blocks | index block | footer block.  A block is one complete gzip member
(zipped files) or raw bytes (unzipped, MakeDatabase -Z).  The GPU never sees this: FileCreator.create
hands per-key block bytes (kmg_db_encode) to IndexedOutputFile.set_current_key/write exactly as
MakeAndWriteKey does (KmerFiles/FileCreator.java:385-391), and KmerFile reads blocks back by key.
Zipped bytes are payload-comparable only: they depend on the zlib build behind java.util.zip.Deflater.
"""
import io
import os
import struct
import zlib

HEADER_ZIPPED, HEADER_RAW = 64, 41
INT_COMPRESSOR_ID = "1"          # Compressor.getStringID(IntCompressor.getID())


def gzip_member(data: bytes, level: int) -> bytes:
    """commons-compress GzipCompressorOutputStream with GzipParameters(level): header
    1f 8b 08 00 | mtime 0 | XFL (2 at level 9, 4 at level 1, else 0) | OS 255; raw deflate; CRC32, ISIZE."""
    xfl = 2 if level == 9 else (4 if level == 1 else 0)
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    body = co.compress(data) + co.flush()
    return b"\x1f\x8b\x08\x00\x00\x00\x00\x00" + bytes([xfl, 0xFF]) + body + \
        struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data) & 0xFFFFFFFF)


def gunzip_members(data: bytes) -> bytes:
    """All concatenated gzip members of a byte string (java.util.zip.GZIPInputStream semantics)."""
    out = []
    while data:
        d = zlib.decompressobj(31)
        out.append(d.decompress(data))
        out.append(d.flush())
        data = d.unused_data
    return b"".join(out)


class IndexedOutputFile:
    """jar IndexedFiles.IndexedOutputFile<Integer> with an IntCompressor key (A.1)."""

    def __init__(self, path, hr=False, zip_level=5):
        """zip_level None = unzipped (MakeDatabase -Z), else the -z level (default 5)."""
        self.path = path
        self.hr = hr
        self.level = zip_level
        self.f = open(path, "wb")
        self.f.write(b"\0" * (HEADER_RAW if zip_level is None else HEADER_ZIPPED))
        self.index = {}                  # key -> [(pos, length)]
        self.cur_key = None
        self.buf = io.BytesIO()

    def _close_block(self):
        if self.cur_key is None:
            return
        data = self.buf.getvalue()
        blob = data if self.level is None else gzip_member(data, self.level)
        pos = self.f.tell()
        self.f.write(blob)
        self.index.setdefault(self.cur_key, []).append((pos, len(blob)))
        self.cur_key = None
        self.buf = io.BytesIO()

    def set_current_key(self, key):
        self._close_block()
        self.cur_key = key

    def write(self, data: bytes):
        self.buf.write(data)

    def write_all(self, data: bytes, key):
        """IndexedOutputFileSet.writeAll: a block of its own (the meta block under key -1)."""
        self.set_current_key(key)
        self.write(data)
        self._close_block()

    def close(self):
        self._close_block()
        # index block, at the file's zip level
        if not self.hr:
            idx = struct.pack(">i", len(self.index))
            for key in sorted(self.index):
                blocks = self.index[key]
                idx += struct.pack(">ii", key, len(blocks))
                for pos, ln in blocks:
                    idx += struct.pack(">qq", pos, ln)
        else:
            idx = "".join("%d\t%s\n" % (key, " ".join("%d:%d" % b for b in self.index[key]))
                          for key in sorted(self.index)).encode()
        blob = idx if self.level is None else gzip_member(idx, self.level)
        ipos = self.f.tell()
        self.f.write(blob)
        # footer block: always level 0 / raw
        footer = ("%-19d\n%-19d\n%s\n%1d\n" % (ipos, len(blob), INT_COMPRESSOR_ID, 1 if self.hr else 0)).encode()
        fblob = footer if self.level is None else gzip_member(footer, 0)
        fpos = self.f.tell()
        self.f.write(fblob)
        # header, written last at offset 0
        header = ("%-19d %-19d%s" % (fpos, len(fblob), " \n")).encode()
        assert len(header) == HEADER_RAW
        hblob = header if self.level is None else gzip_member(header, 0)
        assert len(hblob) == (HEADER_RAW if self.level is None else HEADER_ZIPPED)
        self.f.seek(0)
        self.f.write(hblob)
        self.f.close()


class IndexedInputFile:
    """jar IndexedFiles.IndexedInputFile<Integer> (A.1): header -> footer -> index -> blocks by key."""

    def __init__(self, path):
        self.data = open(path, "rb").read()
        self.zipped = self.data[:2] == b"\x1f\x8b"        # Zip.ZipOrNot.isGZipped
        head = self._block(0, 100)[:HEADER_RAW].decode()
        fpos, flen = [int(x) for x in head.split()[:2]]
        lines = self._block(fpos, flen).decode().split("\n")
        ipos, ilen = int(lines[0].strip()), int(lines[1].strip())
        self.hr = lines[3].strip() == "1"
        # key codec by its id string: "1" IntCompressor, "2" StringCompressor ([short length][bytes]; the
        # preprocessed input of MakeDatabase -p)
        self.string_keys = lines[2].strip() == "2"
        raw = self._block(ipos, ilen)
        self.index = {}
        if not self.hr:
            (n,) = struct.unpack_from(">i", raw, 0)
            off = 4
            for _ in range(n):
                if self.string_keys:
                    (kl,) = struct.unpack_from(">h", raw, off)
                    key = raw[off + 2:off + 2 + kl].decode()
                    (nb,) = struct.unpack_from(">i", raw, off + 2 + kl)
                    off += 2 + kl + 4
                else:
                    key, nb = struct.unpack_from(">ii", raw, off)
                    off += 8
                blocks = []
                for _ in range(nb):
                    blocks.append(struct.unpack_from(">qq", raw, off))
                    off += 16
                self.index[key] = blocks
        else:
            for line in raw.decode().splitlines():
                key, rest = line.split("\t")
                self.index[key if self.string_keys else int(key)] = [tuple(int(x) for x in b.split(":")) for b in rest.split()]

    def _block(self, pos, length):
        blob = self.data[pos:pos + length]
        if not self.zipped:
            return blob
        d = zlib.decompressobj(31)          # one member; trailing bytes of a too-long read are ignored
        return d.decompress(blob) + d.flush()

    def max_index(self):
        return max(self.index)

    def read_key(self, key) -> bytes:
        """All blocks of a key, concatenated in write order (MultiBlockInputStream)."""
        return b"".join(self._block(p, l) for p, l in self.index.get(key, []))

    def read_keys(self, first, last):
        """getInputStream(a, b): blocks of every key in [a, b] inclusive."""
        return b"".join(self.read_key(k) for k in sorted(self.index) if first <= k <= last)


def file_set(path):
    """The reference's multi-file convention: `path` if it exists, else path.1, path.2, ...
    (Database/Matcher.java:70-95,111-128)."""
    if os.path.exists(path):
        return [path]
    out, i = [], 1
    while os.path.exists("%s.%d" % (path, i)):
        out.append("%s.%d" % (path, i))
        i += 1
    return out
