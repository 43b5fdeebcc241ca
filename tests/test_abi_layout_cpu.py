"""The ctypes mirrors in gim_amd/_lib.py against the C structs of include/gim_hip.h: a tiny C program (gcc, the header alone -- it is plain C) prints
sizeof and every field's offset, the test compares them with ctypes'.  A field added to one side only (round 5 grew gim_token_emit three times)
would otherwise show up as garbage arguments on the GPU box."""
import ctypes
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "gim_hip.h")


def _c_fields(struct):
    """field names of `typedef struct <struct> { ... }` in declaration order (arrays and multi-declarator lines included)"""
    src = open(HEADER).read()
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), src, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for stmt in body.split(";"):
        stmt = stmt.strip()
        if not stmt:
            continue
        decls = stmt.split(",")
        first = decls[0].strip()
        names.append(re.sub(r"\[.*", "", first.split()[-1].lstrip("*")))
        for d in decls[1:]:
            names.append(re.sub(r"\[.*", "", d.strip().lstrip("*")))
    return names


@pytest.mark.parametrize("struct,mirror", [("gim_conv_args", "ConvArgs"), ("gim_coarse_args", "CoarseArgs"), ("gim_token_emit", "TokenEmit"),
                                           ("gim_lg_assign_args", "LgAssignArgs"), ("gim_copy_segs", "CopySegs")])
def test_ctypes_mirror_matches_the_header(tmp_path, struct, mirror):
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("gcc not found")
    from gim_amd import _lib
    cls = getattr(_lib, mirror)
    c_names = _c_fields(struct)
    py_names = [f[0] for f in cls._fields_]
    assert c_names == py_names, (c_names, py_names)
    prog = ['#include <stdio.h>', '#include <stddef.h>', '#include "gim_hip.h"', "int main(void) {",
            '  printf("%%zu\\n", sizeof(%s));' % struct]
    prog += ['  printf("%%zu\\n", offsetof(%s, %s));' % (struct, n) for n in c_names]
    prog += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(prog))
    exe = tmp_path / "layout"
    subprocess.run([gcc, "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    assert out[0] == ctypes.sizeof(cls), (struct, out[0], ctypes.sizeof(cls))
    for n, off in zip(c_names, out[1:]):
        assert getattr(cls, n).offset == off, (struct, n, off, getattr(cls, n).offset)
