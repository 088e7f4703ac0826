"""CPU: the three descriptions of the C ABI -- include/calodiff.h, the ctypes binding in calodiffusion_amd/engine.py and the
stub INTEGRATION.md shows a reference maintainer -- must agree on struct layouts and argument lists.  (A binder following a
stale INTEGRATION.md hands cd_plan_create a struct of the wrong size; round 2 shipped exactly that.)"""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT
from calodiffusion_amd import engine

HEADER = os.path.join(ROOT, "include", "calodiff.h")
_CTYPE_OF = {"int32_t": C.c_int32, "uint32_t": C.c_uint32, "float": C.c_float, "int": C.c_int}


def _header_text():
    txt = open(HEADER).read()
    return re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)  # comments out


def _header_structs():
    """{name: [(field, ctype)]} parsed from the typedef struct blocks of calodiff.h."""
    out = {}
    txt = _header_text()
    macros = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+(\w+)\s+(-?\d+)\s*$", txt, flags=re.M)}
    for m in re.finditer(r"typedef struct (\w+) \{(.*?)\} \1;", txt, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            ty, names = decl.split(None, 1)
            for nm in names.split(","):
                nm = nm.strip()
                arr = re.match(r"(\w+)\[(\w+)\]", nm)
                if arr:
                    n = arr.group(2)
                    fields.append((arr.group(1), _CTYPE_OF[ty] * (int(n) if n.isdigit() else macros[n])))
                else:
                    fields.append((nm, _CTYPE_OF[ty]))
        out[m.group(1)] = fields
    return out


def _as_struct(fields):
    return type("S", (C.Structure,), {"_fields_": fields})


def _same_layout(a, b):
    sa, sb = (_as_struct(a), _as_struct(b))
    assert [n for n, _ in a] == [n for n, _ in b]
    assert C.sizeof(sa) == C.sizeof(sb)
    for n, _ in a:
        assert getattr(sa, n).offset == getattr(sb, n).offset and getattr(sa, n).size == getattr(sb, n).size, n


def test_engine_structs_match_the_header():
    hs = _header_structs()
    for name in ("CdUnetDesc", "CdLayerMlpDesc", "CdStep", "CdSamplerOp"):
        _same_layout(hs[name], list(getattr(engine, name)._fields_))
    assert engine.CD_ABI_VERSION == int(re.search(r"#define CD_ABI_VERSION (\d+)", open(HEADER).read()).group(1))


def test_integration_md_stub_matches_the_header():
    """Execute the `class Cd...(C.Structure)` blocks of INTEGRATION.md section B and compare them with the header."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    hs = _header_structs()
    found = {}
    for code in re.findall(r"```python\n(.*?)```", md, flags=re.S):
        for m in re.finditer(r"^class (Cd\w+)\(C\.Structure\):.*?\n((?:[ \t]+.*\n)+)", code, flags=re.M):
            ns = {"C": C}
            exec(m.group(0), ns)  # the documented snippet itself
            found[m.group(1)] = ns[m.group(1)]
    assert set(found) == {"CdUnetDesc", "CdLayerMlpDesc"}, found
    for name, cls in found.items():
        _same_layout(hs[name], list(cls._fields_))
        assert C.sizeof(cls) == C.sizeof(getattr(engine, name))
    ver = re.search(r"cd_abi_version\(\) == (\d+)", md)
    assert ver and int(ver.group(1)) == engine.CD_ABI_VERSION


def test_signature_table_matches_the_prototypes():
    """Argument COUNTS and integer/pointer classes of engine._SIGNATURES against the prototypes of calodiff.h."""
    txt = _header_text()
    protos = {}
    for m in re.finditer(r"^\s*(?:const\s+)?[\w\*]+[\s\*]+(cd_\w+)\s*\((.*?)\)\s*;", txt, flags=re.S | re.M):
        args = m.group(2).strip()
        protos[m.group(1)] = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
    assert set(protos) == set(engine._SIGNATURES), set(protos) ^ set(engine._SIGNATURES)
    for name, (_, argtypes) in engine._SIGNATURES.items():
        decl = protos[name]
        assert len(decl) == len(argtypes), (name, decl, argtypes)
        for d, a in zip(decl, argtypes):
            is_ptr_decl = "*" in d or "[" in d
            is_ptr_bind = a in (C.c_void_p, C.c_char_p) or hasattr(a, "contents") or (hasattr(a, "_type_") and not isinstance(a._type_, str))
            assert is_ptr_decl == is_ptr_bind, (name, d, a)
            if not is_ptr_decl:
                want = {"int": C.c_int, "int64_t": C.c_int64, "uint64_t": C.c_uint64, "size_t": C.c_size_t, "float": C.c_float,
                        "double": C.c_double}[d.replace("const ", "").split()[0]]
                assert a is want, (name, d, a)


def test_header_compiles_as_c_with_the_documented_sizes(tmp_path):
    """The header is plain C (a cgo / JNI binder includes it as such) and its struct sizes are the ones the bindings assume."""
    src = tmp_path / "abi.c"
    src.write_text(
        '#include "calodiff.h"\n'
        f"_Static_assert(sizeof(CdUnetDesc) == {C.sizeof(engine.CdUnetDesc)}, \"CdUnetDesc\");\n"
        f"_Static_assert(sizeof(CdLayerMlpDesc) == {C.sizeof(engine.CdLayerMlpDesc)}, \"CdLayerMlpDesc\");\n"
        f"_Static_assert(sizeof(CdStep) == {C.sizeof(engine.CdStep)}, \"CdStep\");\n"
        f"_Static_assert(sizeof(CdSamplerOp) == {C.sizeof(engine.CdSamplerOp)}, \"CdSamplerOp\");\n"
        f"_Static_assert(CD_ABI_VERSION == {engine.CD_ABI_VERSION}, \"version\");\n"
        "int main(void) { return 0; }\n")
    r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o",
                        str(tmp_path / "abi.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_library_reports_the_abi_version_and_refuses_foreign_structs():
    lib = engine.load_library()
    assert lib.cd_abi_version() == engine.CD_ABI_VERSION
    d = engine.CdUnetDesc()  # struct_size left 0: what a binder of the round-2 header would effectively send
    plan = C.c_void_p()
    assert lib.cd_plan_create(C.byref(d), C.byref(plan)) == -1
    assert b"struct_size" in lib.cd_last_error()


def test_stale_library_is_refused(tmp_path, monkeypatch):
    """load_library compares <lib>.srchash (written by build.py when it links) with the hash of the sources in the tree."""
    from calodiffusion_amd import build
    stamp = engine.LIB_PATH + ".srchash"
    assert open(stamp).read().strip() == build.source_hash()
    monkeypatch.setattr(build, "source_hash", lambda: "0" * 64)
    monkeypatch.delenv("CALODIFF_LIB", raising=False)
    monkeypatch.delenv("CD_SKIP_SRCHASH", raising=False)
    with pytest.raises(RuntimeError, match="was not built from the sources"):
        engine._check_built_from_these_sources()
