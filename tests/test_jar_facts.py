"""SURVEY Appendix A pinned to the one reference artefact that holds it: out/Kmers.jar read AS DATA (tools/jar_facts.py:
zip + class-file constant pools, nothing loaded or executed).  tests/golden/jar_facts.json is the frozen output; where
/root/reference exists it is derived again and compared, and everywhere the constants the oracle's container / codec
code (and the product's host side) rely on are held to it.  This pins the restated constants, not parity."""
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FACTS = json.load(open(os.path.join(ROOT, "tests", "golden", "jar_facts.json")))
JAR = "/root/reference/out/Kmers.jar"


@pytest.mark.skipif(not os.path.exists(JAR), reason="the reference is not on this machine")
def test_frozen_facts_are_what_the_jar_holds():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import jar_facts
    assert jar_facts.facts(JAR) == FACTS


def _imm(cls, method_prefix):
    return {v for m, vals in FACTS[cls]["immediates"].items() if m.startswith(method_prefix) for v in vals}


def test_container_constants():
    """A.1: header / footer formats, fixed header sizes, write buffer, gzip sniffing, gzip member OS byte."""
    out = FACTS["IndexedFiles/IndexedOutputFile"]
    assert "%-19d %-19d%s" in out["strings"] and " \n" in out["strings"] and "%-19d\n%-19d\n%s\n%1d\n" in out["strings"]
    assert 64 in out["int_constants"] and 41 in out["int_constants"]            # seek past the header: zipped / unzipped
    assert 41 in _imm("IndexedFiles/IndexedInputFile", "<init>") and 100 in FACTS["IndexedFiles/IndexedInputFile"]["int_constants"]
    assert " +" in FACTS["IndexedFiles/IndexedInputFile"]["strings"]            # header split
    assert 2000000 in FACTS["IndexedFiles/BlockOutputStream"]["int_constants"]
    assert 35615 in FACTS["Zip/ZipOrNot"]["int_constants"]                      # 0x8b1f little-endian = 1f 8b
    assert 255 in _imm("org/apache/commons/compress/compressors/gzip/GzipParameters", "<init>")   # OS byte
    # ... and the code that relies on them
    slow = open(os.path.join(ROOT, "oracle", "slow_ref.py")).read()
    cont = open(os.path.join(ROOT, "kmers_amd", "container.py")).read()
    cli = open(os.path.join(ROOT, "kmers_amd", "csrc", "kmg_cli.cpp")).read()
    for src in (slow, cont):
        assert '"%-19d %-19d%s"' in src and '"%-19d\\n%-19d\\n%s\\n%1d\\n"' in src
    assert re.search(r"64 if z else 41", slow) and "HEADER_ZIPPED, HEADER_RAW = 64, 41" in cont
    assert "HEADER_ZIPPED = 64, HEADER_RAW = 41" in cli and '"%-19llu %-19llu \\n"' in cli and '"%-19llu\\n%-19llu\\n%s\\n%1d\\n"' in cli
    from kmers_amd import container
    assert container.HEADER_ZIPPED == 64 and container.HEADER_RAW == 41


def test_type_ids_and_text_forms():
    """A.4: type-id arrays and separators of the codecs."""
    ids = {"Compression/SetCompressor": 257, "Compression/ListCompressor": 256, "Compression/MapCompressor": 258,
           "IndexedFiles/IndexCompressor": 1025, "Counts/CountDataType": 2048, "DataTypes/DataPairDataType": 2050,
           "Kmers/KmerWithDataDataType": 2051, "Reads/ReadPosDataType": 2052, "Kmers/SequenceDataType": 2053}
    for cls, want in ids.items():
        assert want in _imm(cls, "getID"), cls
    assert 255 in FACTS["Counts/CountDataType"]["int_constants"]                # the saturation of CountDataType.compress
    assert " " in FACTS["Compression/SetCompressor"]["strings"] and " " in FACTS["Compression/ListCompressor"]["strings"]
    assert {" ", ":"} <= set(FACTS["Compression/MapCompressor"]["strings"])     # entries key:value joined by " "
    assert ":" in FACTS["Reads/ReadPosDataType"]["strings"] and "\t" in FACTS["Kmers/KmerWithDataDataType"]["strings"]
    assert " ~ " in FACTS["DataTypes/DataPairDataType"]["strings"]              # preprocessed text records `taxid ~ SEQUENCE`
    from oracle import slow_ref as R
    assert R.REF_ID == [2048] and R.READ_ID == [257, 2052]
    from kmers_amd import api
    assert api.REF_ID == [2048] and api.READ_ID == [257, 2052]
