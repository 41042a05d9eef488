"""Run a reference config file on the MI355X backend:  python -m volume_renderer_amd.render_config configRenderer.xml

What SceneRenderer.cpp does around the hot path, on Linux and with the integrators on the GPU:
  XMLManager::GetRendererOptions   -> config.load_renderer_options
  SceneLoader::initializeScene     -> objloader.load_obj (+ densefile.load_dense_grid for the grids)
  IntegratorFactory / render       -> vr_backend_render through backend.Backend            (the hot path)
  tonemap, file name, PNG          -> vr_backend_tonemap_device, config.output_file_name, vr_rgb_to_u8 + write_png
                                      (BaseIntegrator.cpp:199-246, :11-30)

Density fields: the reference reads .nvdb with NanoVDB (SceneLoader.cpp:207-230).  This host has no NanoVDB, so a
<densityField> naming X.nvdb is looked up as X.density.vrgrid (+ X.temperature.vrgrid) next to it — the files
host/nvdb_to_dense.cpp writes on the reference side — or given with --density / --temperature, or replaced by the
synthetic cloud of the benchmark with --cloud N R.

Not carried over, on purpose: more than one <model> (the reference's loader overwrites the Embree buffers of the
earlier models but keeps their primitive table entries, SceneLoader.cpp:69-175 — the result is not a meaningful
scene), multiThreaded / freeThreads (CPU scheduling), and the Windows-only console pause.
"""
import argparse
import os
import struct
import sys
import time
import zlib

import numpy as np

from . import abi, backend, config, densefile, objloader

MODES = {"exact": 0, "fast": 1, "exact_skip": 2}


def write_png(path, rgb8):
    """(h, w, 3) uint8, row 0 on top -> 24-bit PNG (what FreeImage_Save(FIF_PNG) is asked for, BaseIntegrator.cpp:14,27)."""
    rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
    h, w, c = rgb8.shape
    assert c == 3

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    raw = np.zeros((h, 1 + 3 * w), dtype=np.uint8)          # filter byte 0 (None) in front of every row
    raw[:, 1:] = rgb8.reshape(h, 3 * w)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(raw.tobytes(), 6)))
        f.write(chunk(b"IEND", b""))


def read_png_rgb8(path):
    """8-bit, non-interlaced RGB or RGBA PNG -> (h, w, 3) uint8, row 0 on top.  Reads what write_png writes and
    what the reference's FreeImage_Save wrote (its committed renders; adaptive row filters 0-4); used by the tests."""
    with open(path, "rb") as f:
        data = f.read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h, bpp = 8, b"", 0, 0, 3
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            w, h, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", body[:13])
            assert depth == 8 and ctype in (2, 6) and interlace == 0, "8-bit non-interlaced RGB(A) only"
            bpp = 3 if ctype == 2 else 4
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    stride = bpp * w
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, 1 + stride)
    out = np.zeros((h, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.int32)
    for y in range(h):
        ft = int(raw[y, 0])
        line = raw[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:                                        # Up
            cur = (line + prev) & 255
        elif ft == 1:                                        # Sub: a prefix sum per channel
            cur = (np.cumsum(line.reshape(w, bpp), axis=0) & 255).reshape(stride)
        else:                                                # Average / Paeth: sequential in x
            cur = np.zeros(stride, dtype=np.int32)
            ln, pv = line.tolist(), prev.tolist()
            acc = [0] * stride
            for i in range(stride):
                a = acc[i - bpp] if i >= bpp else 0
                b = pv[i]
                if ft == 3:
                    pred = (a + b) >> 1
                elif ft == 4:
                    c = pv[i - bpp] if i >= bpp else 0
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                else:
                    raise ValueError(f"PNG filter type {ft}")
                acc[i] = (ln[i] + pred) & 255
            cur = np.array(acc, dtype=np.int32)
        out[y] = cur
        prev = cur
    return out.reshape(h, w, bpp)[:, :, :3].copy()


def find_dense_grids(cfg, density=None, temperature=None):
    """-> (density path or None, temperature path or None) for the config's <densityField> (SceneLoader.cpp:207-209)."""
    if density:
        return density, temperature
    if not cfg.densityField.baseDir:                          # SceneLoader.cpp:207: no baseDir, no grid
        return None, None
    p = cfg.resolve(cfg.densityField)
    if p.endswith(".vrgrid"):
        return p, temperature
    stem = p[:-5] if p.lower().endswith(".nvdb") else p
    d, t = stem + ".density.vrgrid", stem + ".temperature.vrgrid"
    if not os.path.exists(d):
        raise FileNotFoundError(
            f"{p}: this host cannot read .nvdb; convert it on the reference side with host/nvdb_to_dense.cpp "
            f"(expected {d}), or pass --density FILE.vrgrid / --cloud N R")
    return d, temperature or (t if os.path.exists(t) else None)


def load_scene(cfg, be, density=None, temperature=None, cloud=None, log=print):
    """SceneLoader::initializeScene: meshes and grids of the config into the backend."""
    if len(cfg.models) > 1:
        raise ValueError("more than one <model>: not supported (see the module docstring)")
    for m in cfg.models:
        path = cfg.resolve(m)
        base = os.path.dirname(path) + "/"
        model = objloader.load_obj(path, base)
        for w in model.warnings:
            log("Renderer - " + w)
        mesh = model.to_mesh()
        be.set_mesh(mesh)
        log(f"Renderer - {os.path.basename(path)}: {mesh.tris.shape[0]} triangles, {len(model.shapes)} shapes")
    if cloud is not None:
        be.generate_cloud(int(cloud[0]), float(cloud[1]))
        log(f"Renderer - synthetic cloud {int(cloud[0])}^3, radius {float(cloud[1])} voxels")
        return
    d, t = find_dense_grids(cfg, density, temperature)
    if d:
        gd = densefile.load_dense_grid(d, mmap=True)
        gt = densefile.load_dense_grid(t, mmap=True) if t else None
        be.set_density(gd, gt)
        log(f"Renderer - density {os.path.basename(d)} {gd.voxels.shape}" + (f", temperature {os.path.basename(t)}" if t else ""))


def render(cfg, be, log=print):
    """BaseIntegrator::render up to the tonemapped float framebuffer: -> ((h, w, 3) float32, max white, seconds)."""
    import torch
    o = cfg.options
    t0 = time.perf_counter()
    frame = be.render(o, cfg.integrator)
    dt = time.perf_counter() - t0
    log(f"Renderer - Escena rendereada en: {int(dt * 1000)}ms")            # BaseIntegrator.cpp:193
    dev = torch.from_numpy(frame).to(f"cuda:{be.device}")
    white = be.tonemap_device(dev.data_ptr(), o.width * o.height)          # :199-220 on the device copy
    be.synchronize()
    return dev.cpu().numpy(), white, dt


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("config", help="configRenderer.xml (the reference's schema)")
    ap.add_argument("--out-dir", default=".", help="where the PNG goes (the reference writes into its working directory)")
    ap.add_argument("--out", default=None, help="explicit output file name instead of the reference's dated one")
    ap.add_argument("--density", default=None, help=".vrgrid to use instead of the config's <densityField>")
    ap.add_argument("--temperature", default=None, help=".vrgrid of the temperature grid")
    ap.add_argument("--cloud", nargs=2, metavar=("N", "R"), default=None, help="synthetic N^3 cloud of radius R voxels")
    ap.add_argument("--mode", choices=sorted(MODES), default="fast")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args(argv)
    log = (lambda *_: None) if a.quiet else print

    cfg = config.load_renderer_options(a.config)
    with backend.Backend(a.device) as be:                                   # raises when there is no GPU / no library
        be.set_mode(MODES[a.mode])
        load_scene(cfg, be, a.density, a.temperature, a.cloud, log)
        frame, white, _ = render(cfg, be, log)
        stats = be.last_stats()
    rgb8 = backend.rgb_to_u8(frame)
    out = a.out or os.path.join(a.out_dir, config.output_file_name(cfg))
    write_png(out, rgb8)
    log(f"Renderer - Imagen Guardada. {out}  (max white {white:g}, {stats['path_samples']} samples, "
        f"{stats['density_lookups']} lookups)")
    return out


if __name__ == "__main__":
    main(sys.argv[1:])
