#!/usr/bin/env python3
"""Generate tests/golden/ref_kat.npz from the REFERENCE's own code.

Runs only in the build container (needs /root/reference):

    make -C oracle ref && python tests/golden/make_golden.py

oracle/_ref/libvr_refkat.so is the reference's Utils/*.cpp + vendored NanoVDB headers compiled
where they lie (oracle/Makefile, oracle/ref_kat.cpp).  Every array stored here is either an
input chosen below (seeded) or the value the reference code returned for it.  The file is
data only: no reference source text is stored.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SO = os.path.join(ROOT, "oracle", "_ref", "libvr_refkat.so")

f32 = np.float32
pf = C.POINTER(C.c_float)
pi = C.POINTER(C.c_int)


def fp(a):
    return a.ctypes.data_as(pf)


def main():
    L = C.CDLL(SO)
    L.refk_clamp.restype = C.c_float
    L.refk_clamp.argtypes = [C.c_float] * 3
    L.refk_deg2rad.restype = C.c_float
    L.refk_deg2rad.argtypes = [C.c_float]
    L.refk_dot.restype = C.c_float
    L.refk_dot.argtypes = [pf, pf]
    L.refk_cross.argtypes = [pf, pf, pf]
    L.refk_normalize.argtypes = [pf, pf]
    L.refk_rotate.argtypes = [C.c_float] * 3 + [pf]
    L.refk_henyey_greenstein.restype = C.c_float
    L.refk_henyey_greenstein.argtypes = [C.c_float, C.c_float]
    L.refk_get_floats.argtypes = [C.c_uint, C.c_float, C.c_float, C.c_int, pf]
    L.refk_sample_hg.argtypes = [C.c_float, pf, C.c_uint, pf, pf]
    L.refk_perlin_noise.restype = C.c_float
    L.refk_perlin_noise.argtypes = [C.c_float] * 3
    L.refk_perlin_eval_density.restype = C.c_float
    L.refk_perlin_eval_density.argtypes = [pf]
    L.refk_chunks.restype = C.c_int
    L.refk_chunks.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.c_int]
    L.refk_grid_create.restype = C.c_void_p
    L.refk_grid_create.argtypes = [pf, pi, pi, C.c_double, C.POINTER(C.c_double), C.c_float]
    L.refk_grid_info.argtypes = [C.c_void_p, pi, pi, pf, pf, pf, pf]
    L.refk_grid_values.argtypes = [C.c_void_p, pf, C.c_int, pf]
    L.refk_world_ray_clip.restype = C.c_int
    L.refk_world_ray_clip.argtypes = [C.c_void_p, pf, pf, pf]
    L.refk_index_ray_clip.restype = C.c_int
    L.refk_index_ray_clip.argtypes = [C.c_void_p, pf, pf, pf]
    L.refk_ray_at.argtypes = [pf, pf, C.c_float, pf]
    L.refk_grid_destroy.argtypes = [C.c_void_p]

    rng = np.random.default_rng(20261111)
    out = {}

    # ---- Utils::normalize / clamp / deg2rad / dot / cross / rotate (Utils/Utils.cpp) ----
    v = rng.normal(size=(64, 3)).astype(f32) * rng.choice([1e-3, 1.0, 50.0], size=(64, 1)).astype(f32)
    v[0] = 0.0
    r = np.zeros_like(v)
    for i in range(len(v)):
        L.refk_normalize(fp(v[i]), fp(r[i]))
    out["normalize_in"], out["normalize_out"] = v, r

    c = np.array([[0.01, 1.0, 0.5], [0.01, 1.0, 0.001], [0.01, 1.0, 7.0], [0.005, 9999.9, 0.0049], [0, 1, -3],
                  [0, 1, 1.5], [0.01, 1.0, np.inf], [0.01, 1.0, np.nan]], dtype=f32)
    out["clamp_in"] = c
    out["clamp_out"] = np.array([L.refk_clamp(*map(float, row)) for row in c], dtype=f32)

    d = np.array([0, 12.5, 22.5, 30, 37.5, 45, 50, 60, 90, 100, 237], dtype=f32)
    out["deg2rad_in"] = d
    out["deg2rad_out"] = np.array([L.refk_deg2rad(float(x)) for x in d], dtype=f32)

    a = rng.normal(size=(32, 3)).astype(f32)
    b = rng.normal(size=(32, 3)).astype(f32)
    out["dot_a"], out["dot_b"] = a, b
    out["dot_out"] = np.array([L.refk_dot(fp(a[i]), fp(b[i])) for i in range(32)], dtype=f32)
    cr = np.zeros_like(a)
    for i in range(32):
        L.refk_cross(fp(a[i]), fp(b[i]), fp(cr[i]))
    out["cross_out"] = cr

    # rotate(pitch, roll, yaw): shipped cameras (bunny-cloud-configRenderer.xml:37,
    # configRenderer.xml:59, the commented bunny camera) in the call order of
    # BaseIntegrator.cpp:57 (rot.y, rot.x, rot.z) = (yaw attr, pitch attr, roll attr), plus random
    angles = np.array([[-8.0, -19.0, 0.0], [0.0, 20.0, 0.0], [237.0, -15.0, 0.0]] +
                      rng.uniform(-180, 180, size=(13, 3)).tolist(), dtype=f32)
    pts = rng.normal(size=(16, 3)).astype(f32)
    rot = pts.copy()
    for i in range(16):
        L.refk_rotate(float(angles[i, 0]), float(angles[i, 1]), float(angles[i, 2]), fp(rot[i]))
    out["rotate_angles"], out["rotate_in"], out["rotate_out"] = angles, pts, rot

    # ---- PhaseFunction::henyey_greenstein (Utils/PhaseFunction.cpp:4-8) ----
    gs = np.array([-0.2, 0.0, 0.4, 0.924], dtype=f32)
    cs = np.concatenate([np.linspace(-1, 1, 41), rng.uniform(-1.2, 1.2, 23)]).astype(f32)
    hg = np.zeros((len(gs), len(cs)), dtype=f32)
    for i, g in enumerate(gs):
        for j, ct in enumerate(cs):
            hg[i, j] = L.refk_henyey_greenstein(float(g), float(ct))
    out["hg_g"], out["hg_cos"], out["hg_out"] = gs, cs, hg

    # ---- RandomGenerator::getFloat (Utils/RandomGenerator.cpp:4-14) ----
    seeds = np.array([0, 20, 540, 1060, 4294967295], dtype=np.uint32)
    ranges = np.array([[0.00001, 1.0], [0.0, 1.0], [0.0, 0.9999]], dtype=f32)
    gf = np.zeros((len(seeds), len(ranges), 700), dtype=f32)  # > 624 draws: crosses a twist
    for i, s in enumerate(seeds):
        for j, (lo, hi) in enumerate(ranges):
            L.refk_get_floats(int(s), float(lo), float(hi), 700, fp(gf[i, j]))
    out["getfloat_seeds"], out["getfloat_ranges"], out["getfloat_out"] = seeds, ranges, gf

    # ---- DirectionSampler::sampleHenyeyGreenstein (+ONB) (Utils/DirectionSampler.cpp:38-72) ----
    dirs = rng.normal(size=(24, 3)).astype(f32)
    dirs[:8] /= np.linalg.norm(dirs[:8], axis=1, keepdims=True)  # some unit, some not
    dirs[8] = [0, 0, -1]
    dirs[9] = [1, 0, 0]
    dirs[10] = [0, 1, 0]
    dirs[11] = [0.5, 0.2, 0.5]  # |x| == |z| tie in ONB::Update
    hs_out = np.zeros((len(gs), len(dirs), 3), dtype=f32)
    hs_xi = np.zeros((len(gs), len(dirs), 2), dtype=f32)
    hs_seed = rng.integers(0, 2 ** 31, size=(len(gs), len(dirs))).astype(np.uint32)
    for i, g in enumerate(gs):
        for j in range(len(dirs)):
            L.refk_sample_hg(float(g), fp(dirs[j]), int(hs_seed[i, j]), fp(hs_out[i, j]), fp(hs_xi[i, j]))
    out["hgs_g"], out["hgs_dirs"], out["hgs_seed"], out["hgs_xi"], out["hgs_out"] = gs, dirs, hs_seed, hs_xi, hs_out

    # ---- Perlin (Utils/PerlinNoiseSampler.cpp) ----
    pp = rng.uniform(-300, 300, size=(256, 3)).astype(f32)
    pp[:16] = np.round(pp[:16])  # lattice points
    out["perlin_in"] = pp
    out["perlin_noise"] = np.array([L.refk_perlin_noise(*map(float, q)) for q in pp], dtype=f32)
    out["perlin_density"] = np.array([L.refk_perlin_eval_density(fp(q)) for q in pp], dtype=f32)

    # ---- MultithreadingHelper::tryReservingChunk (Utils/MultithreadingHelper.cpp:10-28) ----
    for name, (size, count) in {"chunks_1080": (20, 54), "chunks_2160": (20, 108), "chunks_ragged": (20, 55)}.items():
        buf = (C.c_uint32 * 256)()
        n = L.refk_chunks(size, count, buf, 256)
        out[name] = np.array(buf[:n], dtype=np.uint32)

    # ---- NanoVDB: getValue(Coord::Floor), worldToIndexF, clip (nanovdb/NanoVDB.h, util/Ray.h) ----
    for gi, (shape, bmin, vs, org) in enumerate([
        ((24, 20, 28), (-12, -10, -14), 1.0, (0.0, 0.0, 0.0)),
        ((17, 33, 9), (3, -40, 100), 0.5, (1.25, -2.0, 7.5)),
    ]):
        dense = rng.uniform(0.05, 1.0, size=shape).astype(f32)
        dense[rng.uniform(size=shape) < 0.3] = 0.0  # inactive (background) voxels inside the box
        dense[0, :, :] = 0.0  # an all-background face: the tree bbox must shrink past it
        bmin_a = np.array(bmin, dtype=np.int32)
        bmax_a = bmin_a + np.array(shape, dtype=np.int32) - 1
        origin = (C.c_double * 3)(*org)
        h = L.refk_grid_create(fp(dense), bmin_a.ctypes.data_as(pi), bmax_a.ctypes.data_as(pi), vs, origin, 0.0)
        rb0 = np.zeros(3, dtype=np.int32)
        rb1 = np.zeros(3, dtype=np.int32)
        mm = np.zeros(2, dtype=f32)
        inv = np.zeros(9, dtype=f32)
        vec = np.zeros(3, dtype=f32)
        bg = np.zeros(1, dtype=f32)
        L.refk_grid_info(h, rb0.ctypes.data_as(pi), rb1.ctypes.data_as(pi), fp(mm), fp(inv), fp(vec), fp(bg))
        # lookups: random (inside and outside), exact integers, half integers, just below integers
        n = 512
        lo = bmin_a.astype(f32) - 3
        hi = bmax_a.astype(f32) + 4
        pos = rng.uniform(lo, hi, size=(n, 3)).astype(f32)
        pos[:64] = np.round(pos[:64])
        pos[64:128] = np.round(pos[64:128]) + 0.5
        pos[128:160] = np.nextafter(np.round(pos[128:160]).astype(f32), f32(-1e9))
        vals = np.zeros(n, dtype=f32)
        L.refk_grid_values(h, fp(pos), n, fp(vals))
        # world rays: from outside toward the box, from inside, axis-aligned, missing
        centre_idx = (bmin_a + bmax_a + 1).astype(np.float64) / 2
        centre_w = centre_idx * vs + np.array(org)
        extent_w = float(np.max(bmax_a - bmin_a + 1)) * vs
        nr = 96
        eyes = (centre_w + rng.normal(size=(nr, 3)) * extent_w * 1.5).astype(f32)
        tgt = (centre_w + rng.uniform(-0.6, 0.6, size=(nr, 3)) * extent_w).astype(f32)
        wdirs = (tgt - eyes).astype(f32)
        wdirs[:48] /= np.linalg.norm(wdirs[:48], axis=1, keepdims=True)  # half unit, half not
        eyes[48:56] = (centre_w + rng.uniform(-0.2, 0.2, size=(8, 3)) * extent_w).astype(f32)  # inside
        for k in range(3):  # axis-aligned (zero components -> inf inverse)
            eyes[56 + k] = (centre_w + np.eye(3)[k] * extent_w * 2).astype(f32)
            wdirs[56 + k] = (-np.eye(3)[k]).astype(f32)
        wdirs[60:72] = -wdirs[60:72]  # pointing away: misses
        w_hit = np.zeros(nr, dtype=np.int32)
        w_out = np.zeros((nr, 8), dtype=f32)
        for k in range(nr):
            w_hit[k] = L.refk_world_ray_clip(h, fp(eyes[k]), fp(wdirs[k]), fp(w_out[k]))
        # index rays (the scatter re-clip): origins inside the bbox, HG-like directions
        ni = 64
        ieyes = rng.uniform(rb0.astype(f32), rb1.astype(f32) + 1, size=(ni, 3)).astype(f32)
        idirs = rng.normal(size=(ni, 3)).astype(f32)
        idirs /= np.linalg.norm(idirs, axis=1, keepdims=True)
        idirs[0] = [0, 1, 0]  # the light direction of the shipped configs
        i_hit = np.zeros(ni, dtype=np.int32)
        i_out = np.zeros((ni, 8), dtype=f32)
        for k in range(ni):
            i_hit[k] = L.refk_index_ray_clip(h, fp(ieyes[k]), fp(idirs[k]), fp(i_out[k]))
        ts = rng.uniform(0, 50, size=ni).astype(f32)
        at = np.zeros((ni, 3), dtype=f32)
        for k in range(ni):
            L.refk_ray_at(fp(ieyes[k]), fp(idirs[k]), float(ts[k]), fp(at[k]))
        L.refk_grid_destroy(h)
        p = f"grid{gi}_"
        out[p + "dense"], out[p + "dense_bmin"] = dense, bmin_a
        out[p + "voxel_size"], out[p + "origin"] = np.array([vs]), np.array(org)
        out[p + "bbox_min"], out[p + "bbox_max"] = rb0, rb1
        out[p + "minmax"], out[p + "inv_mat"], out[p + "vec"], out[p + "background"] = mm, inv, vec, bg
        out[p + "pos"], out[p + "values"] = pos, vals
        out[p + "w_eye"], out[p + "w_dir"], out[p + "w_hit"], out[p + "w_ray"] = eyes, wdirs, w_hit, w_out
        out[p + "i_eye"], out[p + "i_dir"], out[p + "i_hit"], out[p + "i_ray"] = ieyes, idirs, i_hit, i_out
        out[p + "at_t"], out[p + "at_out"] = ts, at

    path = os.path.join(HERE, "ref_kat.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
