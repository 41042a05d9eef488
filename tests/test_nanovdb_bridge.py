"""Reference-side bridge (volume_renderer_amd/host/NanoVdbBridge.hpp) against the reference's own types.

oracle/ref_kat.cpp instantiates the bridge templates on the reference's vendored NanoVDB
(FloatGrid, CoordBBox, ReadAccessor) and tinyobj (attrib_t, shape_t, material_t); that library can
only be built where /root/reference exists (oracle/Makefile `ref`), so these tests skip anywhere
else.  The bridge is host code a maintainer compiles into the reference — no GPU involved.
"""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as ol
import scenes
from volume_renderer_amd.abi import Grid, Mesh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "_ref", "libvr_refkat.so")
MODELS = "/root/reference/VolumeRenderer/Models/"

pf, pi, pu = C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_uint)
f32 = np.float32


def fp(a):
    return a.ctypes.data_as(pf)


@pytest.fixture(scope="module")
def ref():
    if not os.path.exists(SO):
        pytest.skip("oracle/_ref/libvr_refkat.so not built (needs /root/reference: make -C oracle ref)")
    L = C.CDLL(SO)
    if not hasattr(L, "refk_bridge_grid"):
        pytest.skip("oracle/_ref predates the bridge entry points: make -C oracle ref")
    L.refk_grid_create.restype = C.c_void_p
    L.refk_grid_create.argtypes = [pf, pi, pi, C.c_double, C.POINTER(C.c_double), C.c_float]
    L.refk_grid_info.argtypes = [C.c_void_p, pi, pi, pf, pf, pf, pf]
    L.refk_grid_values.argtypes = [C.c_void_p, pf, C.c_int, pf]
    L.refk_grid_destroy.argtypes = [C.c_void_p]
    L.refk_bridge_grid.restype = C.c_longlong
    L.refk_bridge_grid.argtypes = [C.c_void_p, pi, pi, pf, pf, pf, pf, C.c_longlong, pf]
    L.refk_grid_mat.argtypes = [C.c_void_p, pf]
    L.refk_index_to_world.argtypes = [C.c_void_p, pf, C.c_int, pf]
    L.refk_bridge_obj.argtypes = [C.c_char_p, C.c_char_p, pu, pf, pu, pu, pf, pf]
    return L


@pytest.mark.parametrize("shape,bmin,vs,org,background", [
    ((24, 20, 28), (-12, -10, -14), 1.0, (0.0, 0.0, 0.0), 0.0),
    ((17, 33, 9), (3, -40, 100), 0.5, (1.25, -2.0, 7.5), 0.0),
    ((9, 8, 21), (-4, 0, -30), 2.0, (-3.0, 0.5, 0.25), 0.125),     # non-zero background
])
def test_describe_grid_is_what_the_accessor_returns(ref, oracle, shape, bmin, vs, org, background):
    rng = np.random.default_rng(5)
    dense = rng.uniform(0.05, 1.0, size=shape).astype(f32)
    dense[rng.uniform(size=shape) < 0.3] = background          # inactive voxels inside the box
    dense[0, :, :] = background                                 # the tree bbox shrinks past this face
    b0 = np.array(bmin, dtype=np.int32)
    b1 = b0 + np.array(shape, dtype=np.int32) - 1
    h = ref.refk_grid_create(fp(dense), b0.ctypes.data_as(pi), b1.ctypes.data_as(pi), vs, (C.c_double * 3)(*org), background)
    try:
        # what the reference reports for the grid (tree bbox, extrema, map)
        rb0, rb1 = np.zeros(3, np.int32), np.zeros(3, np.int32)
        mm, inv, vec, bg = np.zeros(2, f32), np.zeros(9, f32), np.zeros(3, f32), np.zeros(1, f32)
        ref.refk_grid_info(h, rb0.ctypes.data_as(pi), rb1.ctypes.data_as(pi), fp(mm), fp(inv), fp(vec), fp(bg))
        # the bridge's description of the same grid
        g0, g1 = np.zeros(3, np.int32), np.zeros(3, np.int32)
        ginv, gvec, gbm, gmat, rmat = np.zeros(9, f32), np.zeros(3, f32), np.zeros(2, f32), np.zeros(9, f32), np.zeros(9, f32)
        n = ref.refk_bridge_grid(h, g0.ctypes.data_as(pi), g1.ctypes.data_as(pi), fp(ginv), fp(gvec), fp(gbm), None, 0, fp(gmat))
        dims = g1 - g0 + 1
        assert n == int(np.prod(dims))
        vox = np.zeros(n, dtype=f32)
        assert ref.refk_bridge_grid(h, g0.ctypes.data_as(pi), g1.ctypes.data_as(pi), fp(ginv), fp(gvec), fp(gbm), fp(vox), n, fp(gmat)) == n
        ref.refk_grid_mat(h, fp(rmat))
        assert np.array_equal(gmat.view(np.uint32), rmat.view(np.uint32)) and gmat[0] == f32(vs)
        assert np.array_equal(g0, rb0) and np.array_equal(g1, rb1)
        assert rb0[0] == b0[0] + 1
        assert np.array_equal(ginv.view(np.uint32), inv.view(np.uint32)) and np.array_equal(gvec.view(np.uint32), vec.view(np.uint32))
        assert gbm[0] == bg[0] == f32(background) and gbm[1] == mm[1]
        # voxels, k fastest = the source box cropped to the tree bbox
        lo, hi = rb0 - b0, rb1 - b0 + 1
        crop = dense[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]
        vox = vox.reshape(tuple(dims))
        assert np.array_equal(vox.view(np.uint32), crop.view(np.uint32))
        # and a lookup through that description (oracle = what the GPU path reproduces bit for bit) is the
        # reference's getValue(Coord::Floor(p)) — inside, in holes, outside the bbox
        grid = Grid(vox, bbox_min=g0, inv_mat=ginv, vec=gvec, mat=gmat, background=float(gbm[0]), max_value=float(gbm[1]))
        pos = rng.uniform(b0.astype(f32) - 3, b1.astype(f32) + 4, size=(400, 3)).astype(f32)
        pos[:50] = np.round(pos[:50])
        want = np.zeros(len(pos), dtype=f32)
        ref.refk_grid_values(h, fp(pos), len(pos), fp(want))
        got = np.array([oracle.vro_grid_value_at(C.byref(grid.desc), ol.fptr(np.ascontiguousarray(q))) for q in pos], dtype=f32)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        assert oracle.vro_grid_max(C.byref(grid.desc)) == mm[1]
        # index -> world (the grid marchers' light-ray origin, IntegratorNanoVDBSimple.cpp:41)
        wref, wgot = np.zeros((64, 3), f32), np.zeros((64, 3), f32)
        ipos = np.ascontiguousarray(pos[:64])
        ref.refk_index_to_world(h, fp(ipos), 64, fp(wref))
        oracle.vro_index_to_world(C.byref(grid.desc), fp(ipos), 64, fp(wgot))
        assert np.array_equal(wgot.view(np.uint32), wref.view(np.uint32))
    finally:
        ref.refk_grid_destroy(h)


def _load_obj(ref, name):
    path = MODELS + name
    if not os.path.exists(path):
        pytest.skip("reference Models/ not present")
    counts = (C.c_uint * 3)()
    assert ref.refk_bridge_obj(path.encode(), MODELS.encode(), counts, None, None, None, None, None) == 0
    nv, nt, ns = counts[0], counts[1], counts[2]
    verts, tris = np.zeros((nv, 3), f32), np.zeros((nt, 3), np.uint32)
    p2s, dis, dif = np.zeros(nt, np.uint32), np.zeros(ns, f32), np.zeros((ns, 3), f32)
    assert ref.refk_bridge_obj(path.encode(), MODELS.encode(), counts, fp(verts), tris.ctypes.data_as(pu),
                               p2s.ctypes.data_as(pu), fp(dis), fp(dif)) == 0
    return verts, tris, p2s, dis, dif


def test_describe_mesh_on_the_reference_sphere(ref, oracle):
    """SphereOnly.obj: 162 vertices / 320 triangles, one shape; the marcher then sees a medium."""
    verts, tris, p2s, dis, _dif = _load_obj(ref, "SphereOnly.obj")
    assert verts.shape[0] == 162 and tris.shape[0] == 320
    assert tris.max() == 161 and np.all(p2s == 0) and len(dis) == 1 and dis[0] > 0
    r = np.linalg.norm(verts - verts.mean(axis=0), axis=1)
    assert r.std() / r.mean() < 1e-3                               # it is a sphere
    mesh = Mesh(verts, tris, p2s, dis)
    sc, _keep = ol.make_scene(mesh=mesh)
    c = verts.mean(axis=0)
    o = scenes.options_marcher(24, 24, 1, cameraPosition=(float(c[0]), float(c[1]), float(c[2] + 3.0 * r.mean())))
    img = ol.render(sc, o, 0)
    bgc = np.array(o.backgroundColor, dtype=f32)
    inside = np.any(img != bgc, axis=2)
    assert 0.1 < inside.mean() < 0.9 and np.isfinite(img).all()    # silhouette: medium in the middle, sky around


def test_describe_mesh_keeps_the_global_primitive_table(ref):
    """CornellBox-Sphere.obj (2192 triangles, several shapes): prim_to_shape is the reference's
    `primitives` table — one entry per face, shape ids non-decreasing in file order."""
    verts, tris, p2s, dis, dif = _load_obj(ref, "CornellBox-Sphere.obj")
    assert tris.shape[0] == 2192 and tris.max() < verts.shape[0]
    assert len(dis) == p2s.max() + 1 and len(dis) > 1
    assert np.all(np.diff(p2s.astype(np.int64)) >= 0)
    assert np.all((dis > 0) & (dis <= 1))
    # per-shape diffuse colours (densitySampling's tint): the Cornell box has a red and a green wall
    assert dif.shape == (len(dis), 3) and np.all((dif >= 0) & (dif <= 1)) and len(np.unique(dif, axis=0)) >= 3
