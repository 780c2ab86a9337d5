"""The ctypes bindings of neusky_amd/hip.py against the prototypes of include/neusky_hip.h: same parameter count and the same class of every
parameter (pointer / 32-bit integer / 64-bit integer / float).  A binding that disagrees with its prototype is undefined behaviour on the
host side of a GPU call -- nothing a CPU test executes, and nothing a GPU test is sure to notice.  Also: the ctypes mirrors of the
header's structs have the C structs' sizes and the same offset for every field (compiled here with gcc)."""
import ctypes as C
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "neusky_hip.h")


def _prototypes():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    out = {}
    for m in re.finditer(r"\b(int|int64_t|const char\s*\*)\s+(nsky_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        name, params = m.group(2), " ".join(m.group(3).split())
        out[name] = [] if params in ("", "void") else [p.strip() for p in params.split(",")]
    return out


def _class_of_c(param):
    p = re.sub(r"\b(const|restrict|volatile)\b", " ", param)
    if "*" in p or "[" in p or re.search(r"\bnsky_stream_t\b", p):
        return "pointer"
    if re.search(r"\b(float)\b", p):
        return "float"
    if re.search(r"\b(double)\b", p):
        return "double"
    if re.search(r"\b(int64_t|uint64_t|size_t|long)\b", p):
        return "int64"
    if re.search(r"\b(int32_t|uint32_t|int|unsigned)\b", p):
        return "int32"
    raise AssertionError(f"unclassified C parameter {param!r}")


def _class_of_ctypes(t):
    if t in (C.c_void_p, C.c_char_p) or (isinstance(t, type) and issubclass(t, C._Pointer)):
        return "pointer"
    if t is C.c_float:
        return "float"
    if t is C.c_double:
        return "double"
    if t in (C.c_int64, C.c_uint64, C.c_size_t, C.c_long, C.c_ulong):
        return "int64"
    if t in (C.c_int32, C.c_uint32, C.c_int, C.c_uint):
        return "int32"
    raise AssertionError(f"unclassified ctypes parameter {t!r}")


def test_every_binding_matches_its_prototype():
    from neusky_amd import hip
    protos = _prototypes()
    assert len(protos) >= 80, len(protos)
    bound, problems = 0, []
    for name, params in sorted(protos.items()):
        fn = getattr(hip._lib, name)
        if fn.argtypes is None:
            if params:  # declared, exported, but never given argument types: a call would pass Python ints as C ints
                src = open(os.path.join(ROOT, "neusky_amd", "hip.py")).read()
                if re.search(rf"\b{name}\b", src):
                    problems.append(f"{name}: used by hip.py without argtypes")
            continue
        bound += 1
        want = [_class_of_c(p) for p in params]
        got = [_class_of_ctypes(t) for t in fn.argtypes]
        if want != got:
            problems.append(f"{name}: header {want} != binding {got}")
    assert not problems, "\n".join(problems)
    assert bound >= 75, bound


STRUCTS = {"nsky_gemm_desc": "GemmDesc", "nsky_hashgrid_desc": "HashGridDesc", "nsky_film_net": "FilmNet", "nsky_sdf_net": "SdfNet",
           "nsky_chain_layer": "ChainLayer", "nsky_wgrad_problem": "WgradProblem", "nsky_total_segment": "TotalSegment",
           "nsky_segment": "Segment", "nsky_main_losses_desc": "MainLossesDesc", "nsky_ddf_losses_desc": "DDFLossesDesc",
           "nsky_field_net": "FieldNet"}


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc measures the C structs")
def test_struct_mirrors_have_the_c_sizes(tmp_path):
    from neusky_amd import hip
    header = open(HEADER).read()
    names = [c for c in STRUCTS if re.search(rf"\b{c}\b", header) and hasattr(hip, STRUCTS[c])]
    assert len(names) >= 8, names
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "neusky_hip.h"\nint main(void) {\n'
                   + "".join(f'  printf("{c} %zu\\n", sizeof({c}));\n' for c in names) + "  return 0;\n}\n")
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True, capture_output=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    sizes = dict(line.split() for line in out.splitlines())
    wrong = {c: (int(sizes[c]), C.sizeof(getattr(hip, STRUCTS[c]))) for c in names if int(sizes[c]) != C.sizeof(getattr(hip, STRUCTS[c]))}
    assert not wrong, wrong


def _c_struct_fields():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    out = {}
    for m in re.finditer(r"typedef\s+struct\s*\w*\s*\{(.*?)\}\s*(nsky_\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(1).split(";"):
            for part in decl.strip().split(","):
                mm = re.search(r"(\w+)\s*(\[[^\]]*\])?\s*$", part.strip())
                if part.strip() and mm:
                    fields.append(mm.group(1))
        out[m.group(2)] = fields
    return out


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc measures the C structs")
def test_struct_mirrors_have_the_c_field_offsets(tmp_path):
    from neusky_amd import hip
    c_fields = _c_struct_fields()
    lines, checked = [], 0
    for c, py in STRUCTS.items():
        assert c in c_fields, c
        mirror = getattr(hip, py)
        py_fields = [f[0] for f in mirror._fields_]
        assert py_fields == c_fields[c], (c, py_fields, c_fields[c])  # same fields, same order, same names
        for f in py_fields:
            lines.append(f'  printf("{c}.{f} %zu\\n", offsetof({c}, {f}));\n')
    src = tmp_path / "offsets.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "neusky_hip.h"\nint main(void) {\n' + "".join(lines) + "  return 0;\n}\n")
    exe = tmp_path / "offsets"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True, capture_output=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    wrong = []
    for line in out.splitlines():
        key, off = line.split()
        c, f = key.split(".")
        checked += 1
        if getattr(getattr(hip, STRUCTS[c]), f).offset != int(off):
            wrong.append((key, int(off), getattr(getattr(hip, STRUCTS[c]), f).offset))
    assert not wrong, wrong
    assert checked >= 140, checked
