#!/usr/bin/env python3
"""Facts about the jar-only classes of the reference, read from out/Kmers.jar AS DATA.

About half of the hot path's behaviour lives in packages whose source is not in the reference repository
(IndexedFiles, Compression, CountMaps, Streams, Zip ...: SURVEY.md Appendix A); the oracle's container and codec code
relies on constants recovered from their bytecode.  This tool makes that recovery repeatable: it opens the jar as a
zip archive, parses the class files' constant pools and method bodies (JVMS 4: nothing is loaded, linked or executed --
there is no JVM here anyway) and reports, per class, the string / integer constants and the small immediates (bipush /
sipush operands) found in named methods.  `python tools/jar_facts.py [jar] > tests/golden/jar_facts.json` froze the
committed copy; tests/test_jar_facts.py re-derives it wherever /root/reference exists and, everywhere, holds the
oracle's constants to it.  It pins Appendix A, not parity: only outputs of the reference could do that.
"""
import json
import struct
import sys
import zipfile

DEFAULT_JAR = "/root/reference/out/Kmers.jar"

# class -> methods whose immediates are reported (None = every method)
WANTED = {
    "IndexedFiles/IndexedOutputFile": None,
    "IndexedFiles/IndexedInputFile": None,
    "IndexedFiles/BlockOutputStream": None,
    "IndexedFiles/IndexCompressor": None,
    "IndexedFiles/IntegerIndexedOutputFileCache": None,
    "Compression/IntCompressor": None,
    "Compression/SetCompressor": None,
    "Compression/ListCompressor": None,
    "Compression/MapCompressor": None,
    "Compression/StringCompressor": None,
    "Compression/Compressor": None,
    "Zip/ZipOrNot": None,
    "org/apache/commons/compress/compressors/gzip/GzipParameters": None,
    "Counts/CountDataType": None,
    "Reads/ReadPosDataType": None,
    "Kmers/SequenceDataType": None,
    "Kmers/KmerWithDataDataType": None,
    "DataTypes/DataPairDataType": None,
}


def parse_class(data):
    """-> (constant pool dict index -> (tag, value), methods {name+descriptor: code bytes})."""
    if data[:4] != b"\xca\xfe\xba\xbe":
        raise ValueError("not a class file")
    off = 8
    (n,) = struct.unpack_from(">H", data, off)
    off += 2
    cp = {}
    i = 1
    while i < n:
        tag = data[off]
        off += 1
        if tag == 1:                                   # Utf8
            (ln,) = struct.unpack_from(">H", data, off)
            cp[i] = (tag, data[off + 2:off + 2 + ln].decode("utf-8", "replace"))
            off += 2 + ln
        elif tag == 3:                                 # Integer
            cp[i] = (tag, struct.unpack_from(">i", data, off)[0]); off += 4
        elif tag == 4:                                 # Float
            cp[i] = (tag, struct.unpack_from(">f", data, off)[0]); off += 4
        elif tag == 5:                                 # Long (two slots)
            cp[i] = (tag, struct.unpack_from(">q", data, off)[0]); off += 8; i += 1
        elif tag == 6:                                 # Double (two slots)
            cp[i] = (tag, struct.unpack_from(">d", data, off)[0]); off += 8; i += 1
        elif tag in (7, 8, 16, 19, 20):                # Class, String, MethodType, Module, Package
            cp[i] = (tag, struct.unpack_from(">H", data, off)[0]); off += 2
        elif tag in (9, 10, 11, 12, 17, 18):           # Field/Method/InterfaceMethod ref, NameAndType, Dynamic, InvokeDynamic
            cp[i] = (tag, struct.unpack_from(">HH", data, off)); off += 4
        elif tag == 15:                                # MethodHandle
            cp[i] = (tag, struct.unpack_from(">BH", data, off)); off += 3
        else:
            raise ValueError("unknown constant pool tag %d" % tag)
        i += 1
    off += 6                                           # access, this, super
    (ni,) = struct.unpack_from(">H", data, off)
    off += 2 + 2 * ni

    def skip_attrs(o):
        (na,) = struct.unpack_from(">H", data, o)
        o += 2
        attrs = []
        for _ in range(na):
            name_i, ln = struct.unpack_from(">HI", data, o)
            attrs.append((cp[name_i][1], data[o + 6:o + 6 + ln]))
            o += 6 + ln
        return o, attrs

    (nf,) = struct.unpack_from(">H", data, off)
    off += 2
    for _ in range(nf):
        off += 6
        off, _a = skip_attrs(off)
    (nm,) = struct.unpack_from(">H", data, off)
    off += 2
    methods = {}
    for _ in range(nm):
        _acc, name_i, desc_i = struct.unpack_from(">HHH", data, off)
        off += 6
        off, attrs = skip_attrs(off)
        for an, body in attrs:
            if an == "Code":
                (cl,) = struct.unpack_from(">I", body, 4)
                methods[cp[name_i][1] + cp[desc_i][1]] = body[8:8 + cl]
    return cp, methods


def immediates(code):
    """Operands of every bipush (0x10) / sipush (0x11) byte pattern in a method body.  A raw scan, not a disassembly:
    an operand byte of another instruction can look like one, so this answers "does the method push v anywhere" with
    possible false positives, never false negatives."""
    out = set()
    for i, b in enumerate(code):
        if b == 0x10 and i + 1 < len(code):
            out.add(struct.unpack_from(">b", code, i + 1)[0])
        if b == 0x11 and i + 2 < len(code):
            out.add(struct.unpack_from(">h", code, i + 1)[0])
    return out


def facts(jar_path=DEFAULT_JAR):
    out = {}
    with zipfile.ZipFile(jar_path) as z:
        names = set(z.namelist())
        for cls in sorted(WANTED):
            entry = cls + ".class"
            if entry not in names:
                out[cls] = None
                continue
            cp, methods = parse_class(z.read(entry))
            strings = sorted({cp[v][1] for t, v in cp.values() if t == 8})
            ints = sorted({v for t, v in cp.values() if t in (3, 5)})
            imm = {}
            for m, code in sorted(methods.items()):
                vals = sorted(v for v in immediates(code) if v >= 32 or v == -1)
                if vals:
                    imm[m] = vals
            out[cls] = {"strings": strings, "int_constants": ints, "immediates": imm}
    return out


if __name__ == "__main__":
    json.dump(facts(sys.argv[1] if len(sys.argv) > 1 else DEFAULT_JAR), sys.stdout, indent=1, sort_keys=True)
    sys.stdout.write("\n")
