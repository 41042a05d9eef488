"""configRenderer.xml -> Options: the reference's config schema read on Linux.

Follows Utils/XMLManager.cpp:8-110 node by node (same element and attribute names, same defaults) and
pugixml 1.9's text conversions, which are lenient where Python's are strict:

  as_float   strtod on the text: the longest numeric prefix, 0 when there is none      (pugixml.cpp get_value_float)
  as_int     optional sign, "0x" selects base 16, digits until the first non-digit,
             clamped to the type's range                                                (string_to_integer)
  as_uint    the same, negative values give 0
  as_bool    true when the FIRST character is one of 1 t T y Y                           (get_value_bool)
  default    used only when the node / attribute is absent or has no text; an element
             whose text is white space only has no text (parse_default drops it)

Differences, on purpose: the reference always opens "configRenderer.xml" in the working directory
and carries on with all defaults when it is missing (XMLManager.cpp:13 ignores the parse result);
here the path is an argument and a missing or malformed file raises.  baseDir values are Windows
paths ("..\\Models\\"); `resolve` turns the separators and makes them relative to the config file,
which is where the reference's working directory is when it runs.
"""
import os
import re
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field

import numpy as np

from .abi import OptionsPod

_FLOAT_PREFIX = re.compile(r"\s*([+-]?(?:(?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?|inf(?:inity)?|nan))", re.I)
_HEXFLOAT_PREFIX = re.compile(r"\s*([+-]?0[xX](?:[0-9a-fA-F]+\.?[0-9a-fA-F]*|\.[0-9a-fA-F]+)(?:[pP][+-]?\d+)?)")
_INT_PREFIX = re.compile(r"\s*([+-]?)(0[xX][0-9a-fA-F]*|\d*)")

f32 = np.float32


def as_float(text, default=0.0):
    if text is None:
        return f32(default)
    m = _HEXFLOAT_PREFIX.match(text)                      # strtod reads C99 hex floats too
    if m:
        return f32(float.fromhex(m.group(1)))
    m = _FLOAT_PREFIX.match(text)
    return f32(float(m.group(1))) if m else f32(0.0)      # double -> float, as the assignment to a float member does


def _integer(text, lo, hi):
    m = _INT_PREFIX.match(text)
    sign, digits = m.group(1), m.group(2)
    if digits[:2] in ("0x", "0X"):
        v = int(digits[2:], 16) if len(digits) > 2 else 0
    else:
        v = int(digits) if digits else 0
    if sign == "-":
        v = -v
    return min(max(v, lo), hi)


def as_int(text, default=0):
    return int(default) if text is None else _integer(text, -2**31, 2**31 - 1)


def as_uint(text, default=0):
    return int(default) if text is None else _integer(text, 0, 2**32 - 1)


def as_bool(text, default=False):
    return bool(default) if text is None else text[:1] in ("1", "t", "T", "y", "Y")


def _text(node):
    """xml_node::text(): the first PCDATA child; None when the node is absent or holds white space only."""
    if node is None or node.text is None or node.text.strip(" \t\r\n") == "":
        return None
    return node.text


def _attr(node, name):
    return None if node is None else node.get(name)


@dataclass
class Model:
    """Types.h:80-83"""
    baseDir: str = ""
    fileName: str = ""


@dataclass
class RendererConfig:
    """Everything GetRendererOptions returns: the numeric part as the C-ABI's vr_options_pod, the strings beside it."""
    options: OptionsPod
    integrator: str = "deltaTracking"
    models: list = field(default_factory=list)
    densityField: Model = field(default_factory=Model)
    fileName: str = ""
    multiThreaded: bool = True
    multiThreadedFreeThreads: int = 0
    directory: str = "."

    def resolve(self, model):
        """baseDir + fileName (SceneLoader.cpp:71-72, :208) as a path on this system."""
        p = (model.baseDir + model.fileName).replace("\\", "/")
        return p if os.path.isabs(p) else os.path.normpath(os.path.join(self.directory, p))


def _scaled_colour(dst, node, names=("r", "g", "b")):
    """Vec3f(r, g, b) *= multiplier, in float (XMLManager.cpp:61-67, 82-98)."""
    m = as_float(_attr(node, "multiplier"), 1.0)
    for i, n in enumerate(names):
        dst[i] = f32(as_float(_attr(node, n)) * m)


def parse_renderer_options(xml_text, directory="."):
    """XMLManager::GetRendererOptions on a document given as text."""
    doc = ET.fromstring(xml_text)
    root = doc if doc.tag == "options" else doc.find("options")   # doc.child("options")
    o = OptionsPod()
    cfg = RendererConfig(options=o, directory=directory)

    def child(name):
        return None if root is None else root.find(name)

    models = child("models")
    for m in ([] if models is None else models.findall("model")):                     # :18-28
        cfg.models.append(Model(m.get("baseDir") or "", _text(m) or ""))
    df = child("densityField")                                                        # :30-37
    cfg.densityField = Model(_attr(df, "baseDir") or "", _text(df) or "")

    o.sigma_s = as_float(_text(child("sigma_s")), 0.5)                                # :39-40
    o.sigma_a = as_float(_text(child("sigma_a")), 0.5)
    o.rayPerPixelCount = as_uint(_text(child("rayPerPixelCount")), 1)                 # :43
    cfg.integrator = _text(child("integrator")) or "deltaTracking"                    # :44
    o.maxDepth = as_int(_text(child("maxDepth")), 4) & 0xFF                           # :46, uint8_t member
    cfg.fileName = _text(child("fileName")) or ""

    w, h = child("width"), child("height")                                            # :49-54
    o.width = as_int(_text(w), 400) & 0xFFFFFFFF
    o.widthStartOffset = as_int(_attr(w, "startOffset"), 0) & 0xFFFFFFFF
    o.widthReference = as_int(_attr(w, "reference"), 400) & 0xFFFFFFFF
    o.height = as_int(_text(h), 400) & 0xFFFFFFFF
    o.heightStartOffset = as_int(_attr(h, "startOffset"), 0) & 0xFFFFFFFF
    o.heightReference = as_int(_attr(h, "reference"), 400) & 0xFFFFFFFF
    mt = child("multiThreaded")                                                       # :55-57
    cfg.multiThreaded = as_bool(_text(mt), True)
    cfg.multiThreadedFreeThreads = as_int(_attr(mt, "freeThreads"), 0) & 0xFF
    o.multiThreadedChunkSize = as_int(_text(child("multiThreadedChunkSize")), 20) & 0xFFFFFFFF
    o.useImportanceSampling = 1 if as_bool(_text(child("importanceSampling"))) else 0  # :59

    _scaled_colour(o.backgroundColor, child("backgroundColor"))                       # :61-67
    cam = child("camera")                                                             # :69-77
    o.fov = as_float(_attr(cam, "fov"), 90.0)
    for i, n in enumerate(("x", "y", "z")):
        o.cameraPosition[i] = as_float(_attr(cam, n))
    for i, n in enumerate(("pitch", "yaw", "roll")):
        o.cameraRotation[i] = as_float(_attr(cam, n))
    light = child("light")                                                            # :79-86
    for i, n in enumerate(("x", "y", "z")):
        o.lightPosition[i] = as_float(_attr(light, n))
    _scaled_colour(o.lightColor, light)
    _scaled_colour(o.emissionColor, child("emission"))                                # :88-92
    _scaled_colour(o.mediumColor, child("medium"))                                    # :94-98
    o.heyneyGreensteinG = as_float(_text(child("heyneygreenstein_g")))                # :100
    step = child("stepSize")                                                          # :102-104
    o.stepSizeMin = as_float(_attr(step, "min"), 0.01)
    o.stepSizeMax = as_float(_attr(step, "max"), 1.0)
    o.stepSizeMultiplier = as_float(_attr(step, "multiplier"), 1.0)
    o.lightRayDensityMultiplier = as_float(_text(child("lightRayDensityMultiplier")), 1.0)     # :106-107
    o.shadowRayDensityMultiplier = as_float(_text(child("shadowRayDensityMultiplier")), 1.0)
    return cfg


def load_renderer_options(path):
    with open(path, "r", encoding="utf-8", errors="replace") as f:
        text = f.read()
    return parse_renderer_options(text, directory=os.path.dirname(os.path.abspath(path)))


def output_file_name(cfg, when=None):
    """The name BaseIntegrator::render saves under (BaseIntegrator.cpp:225-244):
    "<Mon_dd_hh_mm_ss>_<integrator>_<full|areaX_Y>_ray<spp>_boun<maxDepth>.png" — characters 4..18 of ctime()."""
    import time
    stamp = time.ctime(when)[4:19].replace(":", "_").replace(" ", "_")
    o = cfg.options
    area = o.widthStartOffset > 0 or o.heightStartOffset > 0
    where = f"area{o.widthStartOffset}_{o.heightStartOffset}_" if area else "full_"
    return f"{stamp}_{cfg.integrator}_{where}ray{o.rayPerPixelCount}_boun{o.maxDepth}.png"
