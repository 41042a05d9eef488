"""Config + scene front end: configRenderer.xml -> Options, OBJ/MTL -> vr_mesh_desc.

Three layers of evidence, all on the CPU:
  * tests/golden/frontend.json — what the reference's own tinyobjloader and pugixml produced for the inputs in
    frontend_cases.py (made by tests/golden/make_frontend_golden.py where the reference exists); runs anywhere;
  * the same libraries called live through oracle/_ref (skipped where that is not built) on every model and every
    config file the reference ships;
  * hand-checked values of one config in the reference's schema.
"""
import ctypes as C
import glob
import json
import os
import sys

import numpy as np
import pytest

import frontend_cases as fc
from volume_renderer_amd import config as cfgmod
from volume_renderer_amd import objloader

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "frontend.json")
REF_MODELS = "/root/reference/VolumeRenderer/Models/"
REF_CONFIGS = "/root/reference/VolumeRenderer/SceneRenderer/"
f32 = np.float32


def bits(a):
    return np.ascontiguousarray(a, dtype=f32).view(np.uint32).ravel().tolist()


def mesh_as_lists(model):
    m = model.to_mesh()
    return {"verts": bits(m.verts), "tris": m.tris.ravel().tolist(), "prim_to_shape": m.prim_to_shape.tolist(),
            "dissolve": bits(m.shape_dissolve), "diffuse": bits(m.shape_diffuse)}


@pytest.fixture(scope="module")
def golden():
    with open(GOLDEN) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def ref():
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_frontend_golden as mk
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libvr_refkat.so")):
        pytest.skip("oracle/_ref/libvr_refkat.so not built (needs /root/reference: make -C oracle ref)")
    L = mk.load_ref()
    if not hasattr(L, "refk_xml_number"):
        pytest.skip("oracle/_ref predates the pugixml entry points: make -C oracle ref")
    return L, mk


# ---------------------------------------------------------------- OBJ / MTL

@pytest.mark.parametrize("name", sorted(fc.OBJ_CASES))
def test_obj_cases_match_tinyobj_golden(golden, tmp_path, name):
    path = fc.write_case(str(tmp_path), name, fc.OBJ_CASES[name])
    got = mesh_as_lists(objloader.load_obj(path, str(tmp_path) + "/"))
    assert got == golden["obj"][name]


def test_obj_case_details(tmp_path):
    """What the cases are there to show, stated directly."""
    load = lambda n: objloader.load_obj(fc.write_case(str(tmp_path), n, fc.OBJ_CASES[n]), str(tmp_path) + "/")  # noqa: E731
    cube = load("cube_two_materials")
    assert len(cube.shapes) == 1 and cube.shapes[0].material_ids == [0] * 6 + [1] * 6
    m = cube.to_mesh()
    assert m.shape_dissolve.tolist() == [0.5] and cube.materials[1].dissolve == f32(1.0) - f32(0.25)
    odd = load("odd_numbers")
    assert odd.vertices[0].tolist() == [0.5, -2.5, 1.0]
    assert odd.vertices[2].tolist() == [0.0, 7.0, 10.0]          # "1e" is not a number -> default
    assert odd.vertices[3].tolist() == [6.0, 5.0, 0.0]           # missing z
    dropped = load("object_after_usemtl_drops_shape")
    assert [s.name for s in dropped.shapes] == ["second"] and dropped.to_mesh().shape_dissolve[0] == f32(0.6)
    poly = load("relative_indices_and_polygons")
    assert [len(s.indices) // 3 for s in poly.shapes] == [3, 5] and poly.shapes[1].name == "hexagon part"


def test_obj_errors_are_loud(tmp_path):
    p = tmp_path / "bad.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 0 1 2\n")
    with pytest.raises(objloader.ObjError):                       # LoadObj returns false; the reference exits (SceneLoader.cpp:89)
        objloader.load_obj(str(p))
    with pytest.raises(objloader.ObjError):
        objloader.load_obj(str(tmp_path / "missing.obj"))


def test_parse_double_is_tinyobjs_not_strtods():
    pd = objloader.parse_double
    assert pd("1e") is None and pd("abc") is None and pd("") is None and pd("-") is None
    assert pd("1.") == 1.0 and pd(".5") == 0.5 and pd("-.25e1") == -2.5 and pd("+3") == 3.0
    assert pd("12abc") == 12.0 and pd("1.5x") == 1.5
    # fraction digits are accumulated one by one in double: not always the correctly rounded double
    assert pd("0.1234567891234") == 0.1 + 2 * 0.01 + 3 * 0.001 + 4 * 0.0001 + 5 * 0.00001 + 6 * 0.000001 + 7 * 0.0000001 \
        + 8 * 10.0 ** -8 + 9 * 10.0 ** -9 + 1 * 10.0 ** -10 + 2 * 10.0 ** -11 + 3 * 10.0 ** -12 + 4 * 10.0 ** -13


def test_every_reference_model_matches_tinyobj(ref):
    L, mk = ref
    paths = sorted(glob.glob(REF_MODELS + "*.obj"))
    if not paths:
        pytest.skip("reference Models/ not present")
    assert len(paths) >= 8
    for p in paths:
        want = mk.tinyobj_mesh(L, p, REF_MODELS)
        got = mesh_as_lists(objloader.load_obj(p, REF_MODELS))
        assert want is not None and got == want, os.path.basename(p)


def test_obj_cases_match_tinyobj_live(ref, tmp_path):
    L, mk = ref
    for name, case in fc.OBJ_CASES.items():
        path = fc.write_case(str(tmp_path), name, case)
        assert mesh_as_lists(objloader.load_obj(path, str(tmp_path) + "/")) == mk.tinyobj_mesh(L, path, str(tmp_path) + "/"), name


# ---------------------------------------------------------------- configRenderer.xml

def conversions(text):
    return [float(cfgmod.as_float(text, 7)), float(cfgmod.as_int(text, 7)), float(cfgmod.as_uint(text, 7)),
            1.0 if cfgmod.as_bool(text, True) else 0.0]


def test_text_conversions_match_pugixml_golden(golden):
    for v in fc.XML_VALUES:
        want = golden["xml_values"][v]
        assert conversions(v) == want["attr"], repr(v)                       # an attribute that exists always converts
        as_text = None if v.strip(" \t\r\n") == "" else v                    # white-space-only text is no text at all
        assert conversions(as_text) == want["text"], repr(v)


def test_config_values():
    cfg = cfgmod.parse_renderer_options(fc.XML_CONFIG, directory="/scenes/SceneRenderer")
    o = cfg.options
    assert cfg.integrator == "ratioTracking" and len(cfg.models) == 1 and cfg.models[0].fileName == "SphereOnly.obj"
    assert cfg.resolve(cfg.models[0]) == "/scenes/Models/SphereOnly.obj"
    assert cfg.resolve(cfg.densityField) == "/scenes/DensityFields/cloud.nvdb"
    assert (o.width, o.widthStartOffset, o.widthReference) == (1080, 540, 2160)
    assert (o.height, o.heightStartOffset, o.heightReference) == (2160, 0, 2160)
    assert o.rayPerPixelCount == 16 and o.maxDepth == 16 and o.useImportanceSampling == 0
    assert cfg.multiThreaded and cfg.multiThreadedFreeThreads == 1 and o.multiThreadedChunkSize == 20   # chunk size: default
    assert o.sigma_a == f32(0.01) and o.sigma_s == f32(0.09) and o.heyneyGreensteinG == f32(0.924)
    assert (o.stepSizeMin, o.stepSizeMax, o.stepSizeMultiplier) == (f32(0.005), f32(9999.9), f32(0.001))
    assert list(o.backgroundColor) == [f32(0.36) * f32(4), f32(0.702) * f32(4), f32(0.98) * f32(4)]
    assert list(o.lightColor) == [f32(0.933) * f32(40), f32(0.34) * f32(40), f32(0.16) * f32(40)]
    assert list(o.emissionColor) == [0, 0, 0] and list(o.mediumColor) == [0, 0, 0]            # medium: absent -> 0 * 1
    assert list(o.cameraPosition) == [-10.0, 33.5, 46.0] and list(o.cameraRotation) == [-19.0, -8.0, 0.0] and o.fov == 100.0
    assert o.lightRayDensityMultiplier == 1.0 and o.shadowRayDensityMultiplier == 1.0          # second one: default
    name = cfgmod.output_file_name(cfg, when=0)
    assert name.endswith("_ratioTracking_area540_0_ray16_boun16.png") and len(name.split("_ratioTracking")[0]) == 15


def test_empty_config_gives_the_reference_defaults():
    """XMLManager.cpp defaults = abi.default_options() (which the oracle's vro_options_defaults is checked against)."""
    from volume_renderer_amd import abi
    cfg = cfgmod.parse_renderer_options("<options/>")
    assert bytes(cfg.options) == bytes(abi.default_options())
    assert cfg.integrator == "deltaTracking" and cfg.models == [] and cfg.densityField.baseDir == "" and cfg.multiThreaded


def expected_from_pugixml(L, mk, xml):
    """XMLManager.cpp:39-107 evaluated node by node through the reference's pugixml."""
    from volume_renderer_amd.abi import OptionsPod
    o = OptionsPod()
    for fld, idx, child, attr, kind, default in fc.XML_FIELDS:
        v = mk.xml_number(L, xml, child, attr, kind, default)
        if kind == 1:
            v = int(v) & (0xFF if fld == "maxDepth" else 0xFFFFFFFF)
        elif kind in (2, 3):
            v = int(v)
        if idx is None:
            setattr(o, fld, v)
        else:
            getattr(o, fld)[idx] = v
    for fld, child in fc.XML_COLOURS:
        m = f32(mk.xml_number(L, xml, child, "multiplier", 0, 1.0))
        for i, ch in enumerate("rgb"):
            getattr(o, fld)[i] = f32(mk.xml_number(L, xml, child, ch, 0, 0.0)) * m
    return o


def test_every_reference_config_matches_pugixml(ref):
    L, mk = ref
    paths = sorted(glob.glob(REF_CONFIGS + "*.xml"))
    if not paths:
        pytest.skip("reference SceneRenderer/ not present")
    assert len(paths) >= 10
    buf = C.create_string_buffer(512)
    for p in paths + [None]:
        xml = fc.XML_CONFIG if p is None else open(p, encoding="utf-8", errors="replace").read()
        cfg = cfgmod.parse_renderer_options(xml)
        want = expected_from_pugixml(L, mk, xml)
        for fld, _ in cfg.options._fields_:
            a, b = getattr(cfg.options, fld), getattr(want, fld)
            a, b = (list(a), list(b)) if hasattr(a, "__len__") else (a, b)
            assert a == b, (p, fld, a, b)
        for child, attr, mine in [("integrator", None, cfg.integrator), ("densityField", "baseDir", cfg.densityField.baseDir),
                                  ("densityField", None, cfg.densityField.fileName), ("fileName", None, cfg.fileName)]:
            n = L.refk_xml_string(xml.encode(), child.encode(), attr.encode() if attr else None, 0, buf, 512)
            assert n >= 0
            theirs = buf.value.decode()
            assert mine == (theirs if theirs or child != "integrator" else "deltaTracking"), (p, child)
        for i, m in enumerate(cfg.models):
            assert L.refk_xml_string(xml.encode(), b"model", b"baseDir", i, buf, 512) >= 0 and buf.value.decode() == m.baseDir
            assert L.refk_xml_string(xml.encode(), b"model", None, i, buf, 512) >= 0 and buf.value.decode() == m.fileName
        assert L.refk_xml_string(xml.encode(), b"model", None, len(cfg.models), buf, 512) == -2       # and no further model
