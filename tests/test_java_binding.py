"""The Java side of the boundary, checked without a JDK: every FunctionDescriptor of the FFM binding
(integration/java/Gpu/KmersGpu.java, and the snippet in INTEGRATION.md) against the prototype of the same name in
include/kmers_gpu.h -- arity, argument widths, return type --, every kmg_* symbol the shim sources call against the
binding, and the call-site diff against the reference's sources where those exist."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "kmers_gpu.h")
JAVA_DIR = os.path.join(ROOT, "integration", "java", "Gpu")


def c_prototypes():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = "\n".join(l for l in text.splitlines() if not l.lstrip().startswith("#"))
    protos = {}
    for m in re.finditer(r"(const\s+char\s*\*|void\s*\*|int|void)\s*(kmg_\w+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        params = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        protos[name] = (layout_of(ret, is_return=True), [layout_of(p) for p in params])
    return protos


def layout_of(decl, is_return=False):
    """C declaration of one parameter (or a return type) -> the FFM value layout that carries it."""
    if "*" in decl:
        return "ADDRESS"
    t = re.sub(r"\bconst\b", "", decl).split()
    base = t[0] if is_return or len(t) == 1 else " ".join(t[:-1])
    if base == "void":
        return "VOID"
    if base in ("int", "int32_t", "uint32_t", "unsigned int"):
        return "JAVA_INT"
    if base in ("uint64_t", "int64_t", "long long", "unsigned long long", "size_t"):
        return "JAVA_LONG"
    raise AssertionError("no layout for C type %r" % decl)


def java_descriptors(text):
    out = []
    for m in re.finditer(r'fn\(\s*"(kmg_\w+)"\s*,\s*FunctionDescriptor\.(ofVoid|of)\(([^()]*)\)\s*\)', text):
        name, kind, args = m.group(1), m.group(2), [a.strip() for a in m.group(3).split(",") if a.strip()]
        out.append((name, ("VOID", args) if kind == "ofVoid" else (args[0], args[1:])))
    return out


def test_header_parses_to_every_declared_symbol():
    from kmers_amd import _lib
    protos = c_prototypes()
    assert set(protos) == set(_lib.SYMBOLS), set(protos) ^ set(_lib.SYMBOLS)


@pytest.mark.parametrize("source", ["integration/java/Gpu/KmersGpu.java", "INTEGRATION.md"])
def test_function_descriptors_match_the_header(source):
    protos = c_prototypes()
    descs = java_descriptors(open(os.path.join(ROOT, source)).read())
    assert len(descs) >= 10, "no FunctionDescriptor found in %s" % source
    for name, (ret, args) in descs:
        assert name in protos, "%s: %s is not declared in kmers_gpu.h" % (source, name)
        want_ret, want_args = protos[name]
        assert ret == want_ret, "%s: %s returns %s in the header, %s in the binding" % (source, name, want_ret, ret)
        assert args == want_args, "%s: %s(%s) in the header, (%s) in the binding" % (source, name, ", ".join(want_args), ", ".join(args))


def test_shim_sources_call_only_what_the_binding_has():
    binding = open(os.path.join(JAVA_DIR, "KmersGpu.java")).read()
    handles = dict(re.findall(r"MethodHandle\s+(\w+)\s*=\s*fn\(\"(kmg_\w+)\"", binding))
    # every handle is used, and every call(...) / invokeExact names a handle that exists
    for h in handles:
        assert len(re.findall(r"\b%s\b" % h, binding)) >= 2, "handle %s is never called" % h
    for h in re.findall(r"\bcall\((\w+),", binding) + re.findall(r"\b(\w+)\.invokeExact\(", binding):
        assert h in handles, "call through %s, which is no handle of the binding" % h
    # the public methods the shims use exist in the binding with that many arguments
    methods = {m.group(1): len([a for a in m.group(2).split(",") if a.strip()])
               for m in re.finditer(r"public\s+[\w.<>\[\], ]+?\s+(\w+)\(([^)]*)\)", binding)}
    for f in os.listdir(JAVA_DIR):
        if f == "KmersGpu.java" or not f.endswith(".java"):
            continue
        text = open(os.path.join(JAVA_DIR, f)).read()
        for m in re.finditer(r"\bgpu\.(\w+)\(", text):
            name = m.group(1)
            assert name in methods, "%s calls gpu.%s, which KmersGpu does not have" % (f, name)
            depth, i, nargs, seen = 1, m.end(), 0, False
            while depth:
                ch = text[i]
                if ch in "([{":
                    depth += 1
                elif ch in ")]}":
                    depth -= 1
                elif ch == "," and depth == 1:
                    nargs += 1
                if not ch.isspace() and depth:
                    seen = True
                i += 1
            nargs = nargs + 1 if seen else 0
            assert nargs == methods[name], "%s: gpu.%s called with %d arguments, declared with %d" % (f, name, nargs, methods[name])


def test_callsite_diff_applies_to_the_reference(tmp_path):
    ref = "/root/reference/src"
    if not os.path.isdir(ref) or not shutil.which("patch"):
        pytest.skip("the reference's sources are not on this box")
    shutil.copytree(ref, str(tmp_path / "src"))
    r = subprocess.run(["patch", "-p1", "--dry-run", "-i", os.path.join(ROOT, "integration", "java", "callsites.diff")],
                       cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "FAILED" not in r.stdout and "fuzz" not in r.stdout, r.stdout
