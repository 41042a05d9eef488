"""Wavefront OBJ / MTL -> the triangle description the marchers take (vr_mesh_desc).

The reference loads its models with tinyobjloader 2.0.0-rc (vendored header, `tinyobj::LoadObj` with
triangulation on, SceneLoader.cpp:79) and hands Embree the vertex array plus the flattened vertex
indices of every shape (SceneLoader.cpp:98-175).  Which triangles come out — their order, the shape
they belong to, how a quad is split, what a malformed number parses to — is decided by that loader,
so this module restates the parts of it the path depends on:

  numbers        tiny_obj_loader.h:805-941   a digit-by-digit double accumulation (NOT correctly rounded:
                                             fraction digits are added as digit * 10^-k), cast to float
  indices        :739-760, :1071-1122        1-based, negative = relative to the vertices read so far, 0 = error
  shapes         :2308-2333, :2386-2458,     `usemtl` flushes the pending faces into the current shape with the OLD
                 :2579-2588                  material; `g` / `o` close the shape (with `o` a shape whose faces were all
                                             flushed already is dropped: the loader only keeps it when faces are pending)
  triangulation  :1342-1564                  ear clipping in the plane of the first non-degenerate corner, float arithmetic
  materials      :1639-2012                  newmtl / Kd / Ke / d / Tr (`d` wins over `Tr`, Tr means 1 - dissolve)

`tests/test_frontend.py` compares the result with tinyobj itself (compiled from the reference's sources into
oracle/_ref) on every model the reference ships, bit for bit.  Anything the integrators never read (normals,
texture coordinates, smoothing groups, tags, textures) is parsed only as far as needed to stay in step.
"""
import math
import os
from dataclasses import dataclass, field

import numpy as np

from .abi import Mesh

f32 = np.float32
_WS = " \t"
_POW_LUT = (1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001)


class ObjError(ValueError):
    pass


def parse_double(tok):
    """tiny_obj_loader.h:805-931 — value of `tok` (no white space inside) or None when it is not a number."""
    n, i = len(tok), 0
    if n == 0:
        return None
    sign, dot_first = 1.0, False
    c = tok[0]
    if c in "+-":
        sign = -1.0 if c == "-" else 1.0
        i = 1
        dot_first = i < n and tok[i] == "."
    elif c.isascii() and c.isdigit():
        pass
    elif c == ".":
        dot_first = True
    else:
        return None
    mant, read = 0.0, 0
    if not dot_first:
        while i < n and tok[i] in "0123456789":
            mant = mant * 10 + (ord(tok[i]) - 48)
            i += 1
            read += 1
        if read == 0:
            return None
    exponent = 0
    if i < n and tok[i] == ".":
        i += 1
        read = 1
        while i < n and tok[i] in "0123456789":
            mant += (ord(tok[i]) - 48) * (_POW_LUT[read] if read < len(_POW_LUT) else math.pow(10.0, -read))
            read += 1
            i += 1
    elif i < n and tok[i] in "eE":
        pass
    else:
        return sign * mant
    if i < n and tok[i] in "eE":
        i += 1
        esign = 1
        if i < n and tok[i] in "+-":
            esign = -1 if tok[i] == "-" else 1
            i += 1
        elif i < n and tok[i] in "0123456789":
            pass
        else:
            return None                                   # "1e": an empty exponent is a parse failure
        read = 0
        while i < n and tok[i] in "0123456789":
            exponent = exponent * 10 + (ord(tok[i]) - 48)
            i += 1
            read += 1
        exponent *= esign
        if read == 0:
            return None
    if exponent:
        try:
            return sign * math.ldexp(mant * math.pow(5.0, exponent), exponent)
        except OverflowError:
            return sign * math.inf
    return sign * mant


def _atoi(s):
    """C atoi: leading white space, optional sign, digits; 0 when there are none."""
    i, n = 0, len(s)
    while i < n and s[i] in " \t\n\v\f\r":
        i += 1
    neg = False
    if i < n and s[i] in "+-":
        neg = s[i] == "-"
        i += 1
    v = 0
    while i < n and s[i] in "0123456789":
        v = v * 10 + (ord(s[i]) - 48)
        i += 1
    return -v if neg else v


def _tokens(rest):
    """Split on the characters tinyobj treats as separators inside a line (space, tab, and a stray CR)."""
    return rest.replace("\t", " ").replace("\r", " ").split()


def _reals(rest, count, defaults=(0.0, 0.0, 0.0)):
    """parseReal3 and friends: missing or malformed numbers take the default; each value is cast to float."""
    toks = _tokens(rest)
    out = []
    for k in range(count):
        v = parse_double(toks[k]) if k < len(toks) else None
        out.append(f32(defaults[k] if v is None else v))
    return out


def _fix_index(idx, n):
    if idx > 0:
        return idx - 1
    if idx == 0:
        raise ObjError("index 0")
    return n + idx


@dataclass
class Material:
    """The members of tinyobj::material_t the integrators read (InitMaterial defaults, :1280-1314)."""
    name: str = ""
    diffuse: tuple = (f32(0), f32(0), f32(0))
    emission: tuple = (f32(0), f32(0), f32(0))
    dissolve: np.float32 = f32(1.0)


@dataclass
class Shape:
    name: str = ""
    indices: list = field(default_factory=list)          # vertex indices, three per triangle
    material_ids: list = field(default_factory=list)     # one per triangle


@dataclass
class ObjModel:
    vertices: np.ndarray                                 # (n, 3) float32 — attrib.vertices
    shapes: list
    materials: list
    warnings: list

    def to_mesh(self):
        """What SceneLoader.cpp:98-175 gives Embree plus its `primitives` table and, per shape, the material of the
        shape's first face (:102) — the same description host/NanoVdbBridge.hpp describeMesh produces in C++."""
        tris, p2s, dissolve, diffuse = [], [], [], []
        for s, sh in enumerate(self.shapes):
            tris.extend(sh.indices)
            p2s.extend([s] * (len(sh.indices) // 3))
            mat = sh.material_ids[0] if sh.material_ids else -1
            has = 0 <= mat < len(self.materials)
            dissolve.append(self.materials[mat].dissolve if has else f32(1.0))
            diffuse.append(self.materials[mat].diffuse if has else (f32(0), f32(0), f32(0)))
        return Mesh(self.vertices, np.asarray(tris, dtype=np.uint32).reshape(-1, 3), np.asarray(p2s, dtype=np.uint32),
                    np.asarray(dissolve, dtype=f32), np.asarray(diffuse, dtype=f32).reshape(-1, 3))


def load_mtl(path, materials, material_map, warnings):
    """LoadMtl (:1639-2012) for the members above; appends to `materials` / `material_map`."""
    with open(path, "r", encoding="latin-1", newline="") as f:
        lines = f.read().split("\n")
    cur = Material()
    has_d = False
    for line in lines:
        line = line.rstrip(" \t").rstrip("\r").lstrip(_WS)
        if not line or line[0] == "#":
            continue
        if line.startswith("newmtl") and line[6:7] in (" ", "\t") and len(line) > 6:
            if cur.name:
                material_map.setdefault(cur.name, len(materials))
                materials.append(cur)
            cur = Material(name=line[7:])
            has_d = False
        elif line[:2] == "Kd" and line[2:3] in (" ", "\t") and len(line) > 2:
            cur.diffuse = tuple(_reals(line[2:], 3))
        elif line[:2] == "Ke" and line[2:3] in (" ", "\t") and len(line) > 2:
            cur.emission = tuple(_reals(line[2:], 3))
        elif line[0] == "d" and line[1:2] in (" ", "\t") and len(line) > 1:
            cur.dissolve = _reals(line[1:], 1)[0]
            has_d = True
        elif line[:2] == "Tr" and line[2:3] in (" ", "\t") and len(line) > 2:
            if not has_d:
                cur.dissolve = f32(f32(1.0) - _reals(line[2:], 1)[0])
            else:
                warnings.append(f'both `d` and `Tr` for "{cur.name}": `d` is used')
    material_map.setdefault(cur.name, len(materials))     # the last material is flushed whatever its name (:2006-2009)
    materials.append(cur)


def _pnpoly3(vx, vy, tx, ty):
    """Crossing-number test of (tx, ty) against the triangle (vx, vy), float arithmetic (:1317-1328)."""
    c = False
    j = 2
    for i in range(3):
        if (vy[i] > ty) != (vy[j] > ty):
            with np.errstate(all="ignore"):
                if tx < (vx[j] - vx[i]) * (ty - vy[i]) / (vy[j] - vy[i]) + vx[i]:
                    c = not c
        j = i
    return c


def _triangulate(face, v):
    """One polygon (vertex indices, >= 3 of them) -> list of triangles, as exportGroupsToShape cuts it (:1355-1564).

    `v` is the flat float32 vertex list read so far... the whole file's: faces are exported when their group closes.
    Every polygon, triangles included, goes through the same ear-clipping loop (a triangle falls straight through it).
    """
    nv = len(v)
    n = len(face)
    if n == 3:
        return [tuple(face)]
    ok = lambda vi, a: 0 <= vi * 3 + a < nv                       # noqa: E731  (the loader's bounds checks)
    eps = f32(np.finfo(np.float32).eps)
    axes = [1, 2]
    for k in range(n):                                            # projection plane: first corner with a non-zero cross product
        a, b, c = face[k % n], face[(k + 1) % n], face[(k + 2) % n]
        if not (ok(a, 2) and ok(b, 2) and ok(c, 2)) or a < 0 or b < 0 or c < 0:
            continue
        p0, p1, p2 = v[3 * a:3 * a + 3], v[3 * b:3 * b + 3], v[3 * c:3 * c + 3]
        e0 = [p1[i] - p0[i] for i in range(3)]
        e1 = [p2[i] - p1[i] for i in range(3)]
        cx = abs(e0[1] * e1[2] - e0[2] * e1[1])
        cy = abs(e0[2] * e1[0] - e0[0] * e1[2])
        cz = abs(e0[0] * e1[1] - e0[1] * e1[0])
        if cx > eps or cy > eps or cz > eps:
            if not (cx > cy and cx > cz):
                axes[0] = 0
                if cz > cx and cz > cy:
                    axes[1] = 1
            break
    area = f32(0)
    for k in range(n):
        a, b = face[k % n], face[(k + 1) % n]
        if not (ok(a, axes[0]) and ok(a, axes[1]) and ok(b, axes[0]) and ok(b, axes[1])):
            continue
        ax, ay, bx, by = v[3 * a + axes[0]], v[3 * a + axes[1]], v[3 * b + axes[0]], v[3 * b + axes[1]]
        area = f32(area + (ax * by - ay * bx) * f32(0.5))
    out = []
    rem = list(face)
    guess = 0
    budget = n
    prev = len(rem)
    while len(rem) > 3 and budget > 0:
        m = len(rem)
        if guess >= m:
            guess -= m
        if prev != m:
            prev, budget = m, m
        else:
            budget -= 1
        ind = [rem[(guess + k) % m] for k in range(3)]
        vx = [v[3 * i + axes[0]] if ok(i, axes[0]) and ok(i, axes[1]) else f32(0) for i in ind]
        vy = [v[3 * i + axes[1]] if ok(i, axes[0]) and ok(i, axes[1]) else f32(0) for i in ind]
        cross = (vx[1] - vx[0]) * (vy[2] - vy[1]) - (vy[1] - vy[0]) * (vx[2] - vx[1])
        if cross * area < 0:                                      # a reflex corner: not an ear
            guess += 1
            continue
        overlap = False
        for other in range(3, m):
            o = rem[(guess + other) % m]
            if not (ok(o, axes[0]) and ok(o, axes[1])):
                continue
            if _pnpoly3(vx, vy, v[3 * o + axes[0]], v[3 * o + axes[1]]):
                overlap = True
                break
        if overlap:
            guess += 1
            continue
        out.append(tuple(ind))
        del rem[(guess + 1) % m]
    if len(rem) == 3:
        out.append(tuple(rem))
    return out


def load_obj(path, mtl_basedir=None):
    """tinyobj::LoadObj(&attrib, &shapes, &materials, &warn, &err, path, mtl_basedir) with triangulation on.

    Raises ObjError where LoadObj returns false (the reference then exits, SceneLoader.cpp:89-91)."""
    try:
        with open(path, "r", encoding="latin-1", newline="") as f:
            lines = f.read().split("\n")
    except OSError as e:
        raise ObjError(f"Cannot open file [{path}]") from e
    base = mtl_basedir or ""
    if base and not base.endswith("/"):
        base += "/"
    v = []                                    # flat float32 list, attrib.vertices
    materials, material_map, warnings = [], {}, []
    shapes = []
    shape = Shape()
    pending = []                              # faces of the open group: lists of vertex indices
    pending_other = False                     # `l` / `p` primitives keep a group non-empty (PrimGroup::IsEmpty)
    material = -1
    name = ""

    def flush():
        """exportGroupsToShape: returns whether anything was pending; triangles go to `shape` with the current material."""
        nonlocal pending_other
        if not pending and not pending_other:
            return False
        shape.name = name
        vv = np.asarray(v, dtype=f32)
        for face in pending:
            if len(face) < 3:
                continue
            for tri in _triangulate(face, vv):
                shape.indices.extend(tri)
                shape.material_ids.append(material)
        return True

    for ln, raw in enumerate(lines, 1):
        line = raw[:-1] if raw.endswith("\r") else raw
        line = line.lstrip(_WS)
        if not line or line[0] == "#":
            continue
        c0, c1 = line[0], line[1:2]
        sp = c1 in (" ", "\t") and c1 != ""
        if c0 == "v" and sp:
            v.extend(_reals(line[2:], 3))
        elif c0 == "f" and sp:
            face = []
            for tok in _tokens(line[2:]):
                parts = tok.split("/")                    # i | i/j | i//k | i/j/k  (:1071-1122)
                try:
                    if len(parts) > 3:
                        raise ObjError("too many index fields")
                    face.append(_fix_index(_atoi(parts[0]), len(v) // 3))
                    if len(parts) == 2 or (len(parts) == 3 and parts[1] != ""):
                        _fix_index(_atoi(parts[1]), 1)    # vt / vn are not kept, but a zero there fails the load too
                    if len(parts) == 3:
                        _fix_index(_atoi(parts[2]), 1)
                except ObjError as e:
                    raise ObjError(f"Failed parse `f' line(e.g. zero value for face index. line {ln}.)") from e
            pending.append(face)
        elif (c0 in "lp") and sp:
            for tok in _tokens(line[2:]):
                if _atoi(tok.split("/")[0]) == 0:
                    raise ObjError(f"Failed parse `{c0}' line(e.g. zero value for vertex index. line {ln}.)")
            pending_other = True
        elif line.startswith("usemtl") and line[6:7] in (" ", "\t") and len(line) > 6:
            new = material_map.get(line[7:], -1)
            if new != material:
                flush()
                pending = []
                material = new
        elif line.startswith("mtllib") and line[6:7] in (" ", "\t") and len(line) > 6:
            names = [s for s in line[7:].split(" ") if s]
            if not names:
                warnings.append(f"empty filename for mtllib (line {ln})")
            else:
                for fn in names:
                    p = base + fn
                    if os.path.isfile(p):
                        load_mtl(p, materials, material_map, warnings)
                        break
                    warnings.append(f"Material file [ {p} ] not found.")
                else:
                    warnings.append("Failed to load material file(s). Use default material.")
        elif c0 == "g" and sp:
            flush()
            if shape.indices:
                shapes.append(shape)
            shape = Shape()
            pending, pending_other = [], False
            names = _tokens(line)
            name = " ".join(names[1:]) if len(names) >= 2 else ""
        elif c0 == "o" and sp:
            if flush():
                shapes.append(shape)
            pending, pending_other = [], False
            shape = Shape()
            name = line[2:]
        # vn, vt, s, t and unknown commands: nothing the path reads
    if flush() or shape.indices:
        shapes.append(shape)
    verts = np.asarray(v, dtype=f32).reshape(-1, 3)
    return ObjModel(verts, shapes, materials, warnings)
