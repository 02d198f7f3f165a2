"""The C-ABI library loads and exports every symbol include/kmers_gpu.h declares (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "kmers_gpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kmg_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_exported():
    from kmers_amd import _lib
    names = _declared()
    assert len(names) >= 20
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.SYMBOLS) == names


def test_binding_loads_and_names_kernels():
    from kmers_amd import _lib
    lib = _lib.load()
    n = lib.kmg_prof_count()
    assert n > 5
    names = [lib.kmg_prof_name(i).decode() for i in range(n)]
    assert "sort_scatter_reads" in names and "extract_write" in names


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under kmers_amd/ may reference it."""
    pkg = os.path.join(ROOT, "kmers_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in src.replace("oracle/", "oracle/") or "import oracle" not in src, f
                assert "from oracle" not in src and "import oracle" not in src, f


def test_native_cli_builds_and_refuses_what_is_off_the_gpu_path():
    """kmers_amd/kmg (C++ drivers + container codec): argument handling needs no GPU; options the GPU path does
    not cover exit with status 3 so that a wrapper can fall back to the Java program."""
    import os
    import subprocess
    import kmers_amd
    kmg = os.path.join(os.path.dirname(kmers_amd.__file__), "kmg")
    assert os.path.exists(kmg)
    run = lambda *a: subprocess.run([kmg] + list(a), capture_output=True, text=True, timeout=60)
    assert run().returncode == 1
    # the reference's defaults, -K 32 -k 24, are on the GPU path (tests/test_gpu_programs.py builds with them): here the
    # run gets as far as the missing input / the missing GPU, not status 3
    assert run("Database.MakeDatabase", "-a", "-i", "/nonexistent/x", "-o", "y").returncode == 1
    assert run("Database.MakeDatabase", "-a", "-i", "x", "-o", "y", "-h").returncode == 3    # -h with minK < maxK
    r = run("Database.MakeDatabase", "-a", "-i", "x", "-o", "y", "-K", "20", "-k", "24")
    assert r.returncode == 1 and "above" in r.stderr
    assert run("Database.MakeDatabase", "-a", "-i", "x", "-o", "y", "-K", "31", "-k", "31", "-m", "5").returncode == 1   # -m is on the path: missing input
    assert run("Database.Matcher", "-i", "a", "-d", "b", "-o", "c", "-n", "3").returncode == 1          # any -n runs on the GPU: missing input
    assert run("Database.Matcher", "-i", "a", "-d", "b", "-o", "c", "-x", "t").returncode == 3          # taxonomy: Java path
    r = run("Database.MakeDatabase", "-a", "-q", "-i", "x", "-o", "y", "-K", "31", "-k", "31")
    assert r.returncode == 1 and "exactly one of" in r.stderr


def test_header_is_plain_c(tmp_path):
    """include/kmers_gpu.h is what a cgo / JNI / FFM binding generator reads: it has to stand as C99 on its own."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "h.c"
    src.write_text('#include "kmers_gpu.h"\nint main(void) { return 0; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only",
                        "-I", os.path.join(root, "include"), str(src)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
