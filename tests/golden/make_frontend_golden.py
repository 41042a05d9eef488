"""Generates tests/golden/frontend.json: what the REFERENCE's own tinyobjloader and pugixml (compiled in place
into oracle/_ref/libvr_refkat.so, `make -C oracle ref`; only possible where /root/reference exists) make of the
inputs in tests/frontend_cases.py.  The inputs are ours; the outputs are data.  Floats are stored as uint32 bit
patterns so the comparison is exact.

    python tests/golden/make_frontend_golden.py
"""
import ctypes as C
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import frontend_cases as fc  # noqa: E402

pf, pu = C.POINTER(C.c_float), C.POINTER(C.c_uint)


def load_ref():
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libvr_refkat.so"))
    L.refk_bridge_obj.argtypes = [C.c_char_p, C.c_char_p, pu, pf, pu, pu, pf, pf]
    L.refk_xml_number.restype = C.c_double
    L.refk_xml_number.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_double, C.POINTER(C.c_int)]
    L.refk_xml_string.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int]
    return L


def tinyobj_mesh(L, obj_path, base_dir):
    """-> dict of lists (bit patterns for floats) or None when LoadObj fails."""
    counts = (C.c_uint * 3)()
    if L.refk_bridge_obj(obj_path.encode(), base_dir.encode(), counts, None, None, None, None, None) != 0:
        return None
    nv, nt, ns = counts[0], counts[1], counts[2]
    verts, tris = np.zeros((nv, 3), np.float32), np.zeros((nt, 3), np.uint32)
    p2s, dis, dif = np.zeros(nt, np.uint32), np.zeros(ns, np.float32), np.zeros((ns, 3), np.float32)
    rc = L.refk_bridge_obj(obj_path.encode(), base_dir.encode(), counts, verts.ctypes.data_as(pf), tris.ctypes.data_as(pu),
                           p2s.ctypes.data_as(pu), dis.ctypes.data_as(pf), dif.ctypes.data_as(pf))
    assert rc == 0
    return {"verts": verts.view(np.uint32).ravel().tolist(), "tris": tris.ravel().tolist(), "prim_to_shape": p2s.tolist(),
            "dissolve": dis.view(np.uint32).tolist(), "diffuse": dif.view(np.uint32).ravel().tolist()}


def xml_escape(v):
    return v.replace("&", "&amp;").replace("<", "&lt;").replace('"', "&quot;").replace("\t", "&#9;").replace("\n", "&#10;")


def xml_number(L, xml, child, attr, kind, default):
    present = C.c_int()
    v = L.refk_xml_number(xml.encode(), child.encode(), attr.encode() if attr else None, kind, default, C.byref(present))
    assert present.value == 1, xml
    return v


def main():
    L = load_ref()
    out = {"obj": {}, "xml_values": {}}
    with tempfile.TemporaryDirectory() as d:
        for name, case in fc.OBJ_CASES.items():
            out["obj"][name] = tinyobj_mesh(L, fc.write_case(d, name, case), d + "/")
    for v in fc.XML_VALUES:
        as_text = f"<options><x>{xml_escape(v)}</x></options>"
        as_attr = f'<options><x a="{xml_escape(v)}"/></options>'
        out["xml_values"][v] = {"text": [xml_number(L, as_text, "x", None, k, 7) for k in range(4)],
                                "attr": [xml_number(L, as_attr, "x", "a", k, 7) for k in range(4)]}
    with open(os.path.join(HERE, "frontend.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("wrote", os.path.join(HERE, "frontend.json"))


if __name__ == "__main__":
    main()
