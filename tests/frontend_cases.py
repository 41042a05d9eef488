"""Inputs for the front-end tests (tests/test_frontend.py) and for the script that asks the reference's own
tinyobj / pugixml what they make of them (tests/golden/make_frontend_golden.py).  All text here is ours."""

import os


def write_case(d, name, case):
    """Writes one OBJ case (and its MTL files) into directory d; returns the .obj path."""
    obj, mtls = case
    p = os.path.join(d, name + ".obj")
    with open(p, "w", newline="") as f:
        f.write(obj)
    for fn, text in mtls.items():
        with open(os.path.join(d, fn), "w", newline="") as f:
            f.write(text)
    return p


# name -> (obj text, {mtl file name: text})
OBJ_CASES = {
    # two materials in one group: usemtl flushes faces into the SAME shape with per-face materials; the shape's
    # dissolve is that of its FIRST face (SceneLoader.cpp:102)
    "cube_two_materials": ("""
mtllib cube.mtl
o box
v -2 -1.5 -2
v  2 -1.5 -2
v  2  2.5 -2
v -2  2.5 -2
v -2 -1.5  2
v  2 -1.5  2
v  2  2.5  2
v -2  2.5  2
usemtl fog
f 1 2 3 4
f 5 8 7 6
f 1 5 6 2
usemtl glass
f 2 6 7 3
f 3 7 8 4
f 5 1 4 8
""", {"cube.mtl": """
newmtl fog
Kd 0.25 0.5 0.75
d 0.5
newmtl glass
Kd 1 0.1 0.01
Tr 0.25
"""}),
    # relative (negative) indices, i/j/k forms, a concave pentagon and a hexagon (ear clipping), two `g` groups
    "relative_indices_and_polygons": ("""
v 0 0 0
v 4 0 0
v 4 4 0
v 2 1 0
v 0 4 0
vt 0 0
vn 0 0 1
g concave
f -5/1/1 -4/1/1 -3/1/1 -2/1/1 -1/1/1
v 0 0 1
v 2 0 1
v 3 1.5 1
v 2 3 1
v 0 3 1
v -1 1.5 1
g hexagon part
f 6//1 7//1 8//1 9//1 10//1 11//1
f 6 8 10
""", {}),
    # numbers tinyobj reads its own way: ".5", "-.25e1", "1.", a long fraction (digit-by-digit accumulation), "1e" (malformed
    # -> default 0), a missing z (-> 0), vertex colours after xyz
    "odd_numbers": ("""
v .5 -.25e1 1.
v 0.1234567891234 3.14159265358979 2.718281828459045
v 1e 7 1e+1
v 6 5
v 1.5 2.5 3.5 0.1 0.2 0.3
v 12345.678 -0.000123 1E-3
f 1 2 3
f 4 5 6
""", {}),
    # `o` right after a usemtl change: the faces flushed by usemtl are in a shape that `o` then drops (tiny_obj_loader.h:2441-2449)
    "object_after_usemtl_drops_shape": ("""
mtllib two.mtl
v 0 0 0
v 1 0 0
v 0 1 0
v 1 1 0
usemtl a
f 1 2 3
usemtl b
o second
f 2 4 3
""", {"two.mtl": """
newmtl a
d 0.3
Kd 0.1 0.2 0.3
newmtl b
d 0.6
Kd 0.4 0.5 0.6
Ke 1 1 1
"""}),
    # no materials at all: one unnamed shape, dissolve defaults to 1
    "no_material": ("""
v 0 0 0
v 1 0 0
v 0 1 0
f 1 2 3
""", {}),
}

# values handed to pugixml's as_float / as_int / as_uint / as_bool, as element text and as attribute value
XML_VALUES = ["0.5", " 12", "1e", "1e3", "0x1F", "-3", "+7", "true", "True", "Yes", "yes", "no", "1", "0", "  ", "",
              "1.5e2", ".5", "+.5", "-.5e-1", "abc", "12abc", "9999.9", "999999.9", "0.001", "4294967296", "-2147483649",
              "2147483648", "1.0f", "3,5", "\t2.5\n"]

# a config in the reference's schema exercising defaults (absent nodes), multipliers and crop attributes
XML_CONFIG = """<?xml version="1.0"?>
<options>
  <models>
    <!-- <model baseDir="..\\Models\\">Other.obj</model> -->
    <model baseDir="..\\Models\\">SphereOnly.obj</model>
  </models>
  <densityField baseDir="..\\DensityFields\\">cloud.nvdb</densityField>
  <sigma_a>0.01</sigma_a>
  <sigma_s>0.09</sigma_s>
  <heyneygreenstein_g>0.924</heyneygreenstein_g>
  <rayPerPixelCount>16</rayPerPixelCount>
  <stepSize min="0.005" max="9999.9" multiplier="0.001"/>
  <maxDepth>16</maxDepth>
  <integrator>ratioTracking</integrator>
  <width startOffset="540" reference="2160">1080</width>
  <height startOffset="0" reference="2160">2160</height>
  <multiThreaded freeThreads="1">true</multiThreaded>
  <importanceSampling>false</importanceSampling>
  <camera x="-10.0" y="33.5" z="46.0" pitch="-19.0" yaw="-8.0" roll="0.0" fov="100"/>
  <backgroundColor r="0.36" g="0.702" b="0.98" multiplier="4.0"/>
  <light x="0" y="1" z="0" r="0.933" g="0.34" b="0.16" multiplier="40.0"/>
  <emission r="0.25" g="0.125" b="0.025" multiplier="0.0"/>
  <lightRayDensityMultiplier>1.0</lightRayDensityMultiplier>
</options>
"""

# (OptionsPod field, index or None, child, attribute or None, kind, default) — XMLManager.cpp:39-107; kind as refk_xml_number
XML_FIELDS = [
    ("sigma_s", None, "sigma_s", None, 0, 0.5), ("sigma_a", None, "sigma_a", None, 0, 0.5),
    ("rayPerPixelCount", None, "rayPerPixelCount", None, 2, 1), ("maxDepth", None, "maxDepth", None, 1, 4),
    ("width", None, "width", None, 1, 400), ("widthStartOffset", None, "width", "startOffset", 1, 0),
    ("widthReference", None, "width", "reference", 1, 400),
    ("height", None, "height", None, 1, 400), ("heightStartOffset", None, "height", "startOffset", 1, 0),
    ("heightReference", None, "height", "reference", 1, 400),
    ("multiThreadedChunkSize", None, "multiThreadedChunkSize", None, 1, 20),
    ("useImportanceSampling", None, "importanceSampling", None, 3, 0),
    ("fov", None, "camera", "fov", 0, 90.0),
    ("cameraPosition", 0, "camera", "x", 0, 0), ("cameraPosition", 1, "camera", "y", 0, 0), ("cameraPosition", 2, "camera", "z", 0, 0),
    ("cameraRotation", 0, "camera", "pitch", 0, 0), ("cameraRotation", 1, "camera", "yaw", 0, 0), ("cameraRotation", 2, "camera", "roll", 0, 0),
    ("lightPosition", 0, "light", "x", 0, 0), ("lightPosition", 1, "light", "y", 0, 0), ("lightPosition", 2, "light", "z", 0, 0),
    ("heyneyGreensteinG", None, "heyneygreenstein_g", None, 0, 0),
    ("stepSizeMin", None, "stepSize", "min", 0, 0.01), ("stepSizeMax", None, "stepSize", "max", 0, 1.0),
    ("stepSizeMultiplier", None, "stepSize", "multiplier", 0, 1.0),
    ("lightRayDensityMultiplier", None, "lightRayDensityMultiplier", None, 0, 1.0),
    ("shadowRayDensityMultiplier", None, "shadowRayDensityMultiplier", None, 0, 1.0),
]
# colours: field, child  (r, g, b attributes times the `multiplier` attribute, default 1)
XML_COLOURS = [("backgroundColor", "backgroundColor"), ("lightColor", "light"), ("emissionColor", "emission"), ("mediumColor", "medium")]
