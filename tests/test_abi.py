"""CPU: the C-ABI library builds for gfx950, loads, and exports exactly what include/*.h declares.
No kernel is launched here; only argument validation paths (which return before any launch) are called."""
import ctypes as C
import os
import re

from coma_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = []
    for f in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if f.endswith(".h"):
            txt = open(os.path.join(ROOT, "include", f)).read()
            txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
            names += re.findall(r"\b((?:coma|sd|seg)_[a-z0-9_]+)\s*\(", txt)
    return sorted(set(names))


def test_every_declared_symbol_is_exported_and_bound(hip_lib):
    decl = _declared()
    assert "coma_contact_accumulate_f32" in decl and len(decl) >= 10
    for name in decl:
        assert hasattr(hip_lib, name), f"{name} declared in include/ but not exported"
    assert sorted(_lib.SIGNATURES) == decl, "ctypes signature table out of sync with the header"


def test_abi_version_and_error_text(hip_lib):
    assert hip_lib.coma_abi_version() == 8
    rc = hip_lib.coma_nearest_vertex_i64(None, None, 1, 1, None, None)
    assert rc == -1 and b"null pointer" in hip_lib.coma_last_error()
    one = C.c_void_p(8)   # never dereferenced: size validation fails first
    rc = hip_lib.coma_contact_accumulate_f32(one, one, one, one, 5, one, 1, 2, 3, 4, _lib.vec3([0, 0, 1]),
                                             _lib.vec3([0, 1, 0]), 0.07, 0.03, 0.25, 1e-10, one, one, one, one, one, None)
    assert rc == -1 and b"obj_sample_stride" in hip_lib.coma_last_error()
    rc = hip_lib.coma_occupancy_splat(one, 1, 0, 8, one, 0.3, 0.9, one, None)
    assert rc == -1


def test_library_is_in_tree():
    assert os.path.dirname(_lib.LIB_PATH) == os.path.join(ROOT, "coma_amd")
