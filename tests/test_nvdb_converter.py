""".nvdb -> .vrgrid -> abi.Grid.

host/nvdb_to_dense.cpp needs the reference's vendored NanoVDB headers, so it is built only where the reference
is (oracle/Makefile `ref` -> oracle/_ref/nvdb_to_dense); those tests skip elsewhere.  The .nvdb files are written
here by the reference's own nanovdb::io::writeGrid (the reference ships no .nvdb asset).  The file format itself
(densefile.py <-> DenseGridFile.hpp) is tested everywhere.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ol
from volume_renderer_amd import densefile
from volume_renderer_amd.abi import Grid

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "_ref", "libvr_refkat.so")
TOOL = os.path.join(ROOT, "oracle", "_ref", "nvdb_to_dense")
pf, pi = C.POINTER(C.c_float), C.POINTER(C.c_int)
f32 = np.float32


def test_vrgrid_round_trip(tmp_path):
    rng = np.random.default_rng(3)
    vox = rng.uniform(0, 1, size=(5, 7, 3)).astype(f32)
    inv = [0.5, 0, 0, 0, 0.5, 0, 0, 0, 0.5]
    g = Grid(vox, bbox_min=(-2, 10, 0), inv_mat=inv, vec=(1.5, -2.0, 0.25), mat=[2, 0, 0, 0, 2, 0, 0, 0, 2], background=0.125, max_value=0.75)
    p = str(tmp_path / "g.vrgrid")
    densefile.save_dense_grid(p, g)
    assert os.path.getsize(p) == 136 + vox.size * 4
    for mm in (False, True):
        h = densefile.load_dense_grid(p, mmap=mm)
        assert bytes(h.desc)[8:] == bytes(g.desc)[8:]                      # everything but the voxel pointer
        assert np.array_equal(np.asarray(h.voxels).view(np.uint32), vox.view(np.uint32))
    with open(p, "r+b") as f:
        f.truncate(136 + 8)
    with pytest.raises(ValueError):
        densefile.load_dense_grid(p)
    (tmp_path / "x.vrgrid").write_bytes(b"not a grid")
    with pytest.raises(ValueError):
        densefile.load_dense_grid(str(tmp_path / "x.vrgrid"))


@pytest.fixture(scope="module")
def ref():
    if not (os.path.exists(SO) and os.path.exists(TOOL)):
        pytest.skip("oracle/_ref not built (needs /root/reference: make -C oracle ref)")
    L = C.CDLL(SO)
    if not hasattr(L, "refk_grid_write"):
        pytest.skip("oracle/_ref predates refk_grid_write: make -C oracle ref")
    L.refk_grid_create_named.restype = C.c_void_p
    L.refk_grid_create_named.argtypes = [pf, pi, pi, C.c_double, C.POINTER(C.c_double), C.c_float, C.c_char_p]
    L.refk_grid_info.argtypes = [C.c_void_p, pi, pi, pf, pf, pf, pf]
    L.refk_grid_mat.argtypes = [C.c_void_p, pf]
    L.refk_grid_values.argtypes = [C.c_void_p, pf, C.c_int, pf]
    L.refk_grid_write.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p]
    L.refk_grid_destroy.argtypes = [C.c_void_p]
    return L


def make_ref_grid(L, dense, bmin, vs, org, background, name):
    b0 = np.array(bmin, dtype=np.int32)
    b1 = b0 + np.array(dense.shape, dtype=np.int32) - 1
    return L.refk_grid_create_named(dense.ctypes.data_as(pf), b0.ctypes.data_as(pi), b1.ctypes.data_as(pi), vs,
                                    (C.c_double * 3)(*org), background, name)


def check_against_ref(L, oracle, h, grid):
    rb0, rb1 = np.zeros(3, np.int32), np.zeros(3, np.int32)
    mm, inv, vec, bg, mat = np.zeros(2, f32), np.zeros(9, f32), np.zeros(3, f32), np.zeros(1, f32), np.zeros(9, f32)
    L.refk_grid_info(h, rb0.ctypes.data_as(pi), rb1.ctypes.data_as(pi), mm.ctypes.data_as(pf), inv.ctypes.data_as(pf),
                     vec.ctypes.data_as(pf), bg.ctypes.data_as(pf))
    L.refk_grid_mat(h, mat.ctypes.data_as(pf))
    d = grid.desc
    assert list(d.bbox_min) == rb0.tolist() and list(d.bbox_max) == rb1.tolist()
    assert np.array_equal(np.array(d.inv_mat, dtype=f32).view(np.uint32), inv.view(np.uint32))
    assert np.array_equal(np.array(d.mat, dtype=f32).view(np.uint32), mat.view(np.uint32))
    assert np.array_equal(np.array(d.vec, dtype=f32).view(np.uint32), vec.view(np.uint32))
    assert d.background == bg[0] and d.max_value == mm[1]
    # lookups through the loaded description = the reference accessor's, inside, in holes and outside the bbox
    rng = np.random.default_rng(11)
    pos = rng.uniform(rb0.astype(f32) - 3, rb1.astype(f32) + 4, size=(300, 3)).astype(f32)
    want = np.zeros(len(pos), dtype=f32)
    L.refk_grid_values(h, pos.ctypes.data_as(pf), len(pos), want.ctypes.data_as(pf))
    got = np.array([oracle.vro_grid_value_at(C.byref(d), ol.fptr(np.ascontiguousarray(q))) for q in pos], dtype=f32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_nvdb_with_density_and_temperature(ref, oracle, tmp_path):
    L = ref
    rng = np.random.default_rng(8)
    dens = rng.uniform(0.05, 1.0, size=(19, 12, 27)).astype(f32)
    dens[rng.uniform(size=dens.shape) < 0.4] = 0.0
    temp = rng.uniform(0.0, 3.0, size=(10, 10, 10)).astype(f32)
    hd = make_ref_grid(L, dens, (-9, 4, -13), 0.5, (1.0, -2.0, 0.25), 0.0, b"density")
    ht = make_ref_grid(L, temp, (-5, 6, -5), 0.5, (1.0, -2.0, 0.25), 0.0, b"temperature")
    try:
        nvdb = str(tmp_path / "fire.nvdb")
        assert L.refk_grid_write(nvdb.encode(), ht, hd) == 0               # temperature first: grids are found by NAME
        prefix = str(tmp_path / "fire")
        r = subprocess.run([TOOL, nvdb, prefix], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        check_against_ref(L, oracle, hd, densefile.load_dense_grid(prefix + ".density.vrgrid"))
        check_against_ref(L, oracle, ht, densefile.load_dense_grid(prefix + ".temperature.vrgrid", mmap=True))
    finally:
        L.refk_grid_destroy(hd)
        L.refk_grid_destroy(ht)


def test_nvdb_without_temperature_and_without_density(ref, oracle, tmp_path):
    L = ref
    dens = np.random.default_rng(2).uniform(0.1, 1.0, size=(8, 8, 8)).astype(f32)
    hd = make_ref_grid(L, dens, (0, 0, 0), 1.0, (0.0, 0.0, 0.0), 0.0, b"density")
    ho = make_ref_grid(L, dens, (0, 0, 0), 1.0, (0.0, 0.0, 0.0), 0.0, b"smoke")
    try:
        nvdb = str(tmp_path / "cloud.nvdb")
        assert L.refk_grid_write(nvdb.encode(), hd, None) == 0
        r = subprocess.run([TOOL, nvdb, str(tmp_path / "cloud")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert os.path.exists(tmp_path / "cloud.density.vrgrid") and not os.path.exists(tmp_path / "cloud.temperature.vrgrid")
        check_against_ref(L, oracle, hd, densefile.load_dense_grid(str(tmp_path / "cloud.density.vrgrid")))
        # no grid named "density": the reference throws (SceneLoader.cpp:212-213); the tool reports and fails
        other = str(tmp_path / "other.nvdb")
        assert L.refk_grid_write(other.encode(), ho, None) == 0
        r = subprocess.run([TOOL, other, str(tmp_path / "other")], capture_output=True, text=True)
        assert r.returncode == 1 and "density" in r.stderr
        r = subprocess.run([TOOL, str(tmp_path / "missing.nvdb"), str(tmp_path / "m")], capture_output=True, text=True)
        assert r.returncode == 1
    finally:
        L.refk_grid_destroy(hd)
        L.refk_grid_destroy(ho)
