"""The ".vrgrid" dense grid file (layout: volume_renderer_amd/host/DenseGridFile.hpp).

`host/nvdb_to_dense.cpp`, built on the reference side against its vendored NanoVDB, turns the .nvdb a config's
<densityField> names (SceneLoader.cpp:207-230) into one .vrgrid per grid ("density", "temperature"); this module
loads them as abi.Grid for vr_backend_set_density.  Nothing here reads .nvdb.
"""
import struct

import numpy as np

from .abi import Grid

MAGIC = b"VRGRID01"
_HEADER = struct.Struct("<8s3i3i9f3f9fffIQ")       # 136 bytes
assert _HEADER.size == 136


def save_dense_grid(path, grid):
    """abi.Grid -> file."""
    d = grid.desc
    n = int(np.prod(grid.voxels.shape))
    with open(path, "wb") as f:
        f.write(_HEADER.pack(MAGIC, *d.bbox_min, *d.bbox_max, *d.inv_mat, *d.vec, *d.mat, d.background, d.max_value, 0, n))
        f.write(np.ascontiguousarray(grid.voxels, dtype="<f4").tobytes())


def load_dense_grid(path, mmap=False):
    """file -> abi.Grid.  mmap=True maps the voxels instead of reading them (the upload then streams from the page cache)."""
    with open(path, "rb") as f:
        head = f.read(_HEADER.size)
    if len(head) != _HEADER.size or head[:8] != MAGIC:
        raise ValueError(f"{path}: not a .vrgrid file")
    v = _HEADER.unpack(head)
    bmin, bmax = v[1:4], v[4:7]
    inv_mat, vec, mat = v[7:16], v[16:19], v[19:28]
    background, max_value, n = v[28], v[29], v[31]
    dims = tuple(max(0, bmax[a] - bmin[a] + 1) for a in range(3))
    if n != dims[0] * dims[1] * dims[2]:
        raise ValueError(f"{path}: voxel count {n} does not match the bbox {dims}")
    if mmap:
        vox = np.memmap(path, dtype="<f4", mode="r", offset=_HEADER.size, shape=dims)
    else:
        vox = np.fromfile(path, dtype="<f4", count=n, offset=_HEADER.size)
        if vox.size != n:
            raise ValueError(f"{path}: truncated")
        vox = vox.reshape(dims)
    return Grid(vox, bbox_min=bmin, inv_mat=inv_mat, vec=vec, mat=mat, background=background, max_value=max_value)
