"""The on-disk container (jar IndexedFiles.*): the product's host-side codec
(kmers_amd/container.py) against the oracle's independent restatement, both directions, plus the
hand-derived header vector B17.  No GPU needed."""
import os

import numpy as np
import pytest

from kmers_amd import container, parse
from oracle import slow_ref as R


def test_b17_unzipped_header(tmp_path):
    p = str(tmp_path / "h.db")
    o = container.IndexedOutputFile(p, zip_level=None)
    o.write_all(b"M" * 10, -1)
    o.close()
    raw = open(p, "rb").read()
    head = raw[:41].decode()
    fpos, flen = [int(x) for x in head.split()]
    assert head == "%-19d %-19d \n" % (fpos, flen) and len(head) == 41
    assert raw[fpos:fpos + flen].decode().split("\n")[2:4] == ["1", "0"]


def test_zipped_header_is_a_64_byte_stored_gzip_member(tmp_path):
    p = str(tmp_path / "z.db")
    o = container.IndexedOutputFile(p, zip_level=5)
    o.write_all(b"M", -1)
    o.close()
    raw = open(p, "rb").read()
    assert raw[:10] == bytes.fromhex("1f8b0800000000" + "0000ff")
    assert len(container.gunzip_members(raw[:64])) == 41
    assert container.gzip_member(b"", 5)[:10] == bytes.fromhex("1f8b08000000000000ff") and len(container.gzip_member(b"", 5)) == 20
    assert container.gzip_member(b"x", 9)[8] == 2 and container.gzip_member(b"x", 1)[8] == 4


@pytest.mark.parametrize("zip_level", [None, 5, 1])
def test_product_and_oracle_containers_read_each_other(tmp_path, zip_level):
    blocks = R.build_ref_db(b">7\nACGTNACGTACCGGTTAACG\n>9\nGGGTTTACGT\n", 4, 2)
    a, b = str(tmp_path / "oracle.db"), str(tmp_path / "product.db")
    R.write_indexed_file(a, blocks, zip_level)
    out = container.IndexedOutputFile(b, zip_level=zip_level)
    out.write_all(blocks[-1], -1)
    for key in range(16):
        out.set_current_key(key)
        out.write(blocks[key])
    out.close()
    if zip_level is None:
        assert open(a, "rb").read() == open(b, "rb").read()          # file-byte-exact (tier T2)
    else:
        assert len(open(a, "rb").read()) == len(open(b, "rb").read())
    assert R.read_indexed_file(b) == blocks
    f = container.IndexedInputFile(a)
    assert {k: f.read_key(k) for k in f.index} == blocks
    assert f.max_index() == 15 and f.read_keys(1, 7) == b"".join(blocks[k] for k in range(1, 8))


def test_block_decoder_matches_oracle_decoder():
    fa = b">7\nACGTNACGTACCGGTTAACGACGT\n>-9\nGGGTTTACGTAAACCC\n"
    fq = b"@a\nACGTACGTAC\n+\nIIIIIIIIII\n@b\nACGTACGTTT\n+\nIIIIIIIIII\n"
    for blocks, kind in ((R.build_ref_db(fa, 5, 2), 0), (R.build_read_db(fq, 5, 2)[0], 1)):
        data = b"".join(blocks[k] for k in range(16))
        keys, val_off, va, vb = parse.decode_blocks(data, kind)
        want = []
        for k in range(16):
            want += R.decode_block(blocks[k], "ref" if kind == 0 else "read")
        assert len(keys) == len(want)
        for i, (kmer, d) in enumerate(want):
            assert int(keys[i]) == int.from_bytes(kmer.cbytes().ljust(8, b"\0"), "big")
            got = list(zip(va[int(val_off[i]):int(val_off[i + 1])].tolist(), vb[int(val_off[i]):int(val_off[i + 1])].tolist()))
            assert got == (sorted(d.items()) if kind == 0 else [tuple(x) for x in d])
    with pytest.raises(ValueError):
        parse.decode_blocks(b"\x00\x21" + b"\0" * 20, 0)             # a 33-mer: not representable
