"""Multi-rank row scheduler on CPU: world_size 2 and 3 over gloo.

Each rank renders its interleaved rows (here with the CPU oracle standing in for the HIP
backend — same row-set contract, same Philox streams) and the all_gather'ed frame on rank 0
must equal the single-process frame bit for bit, including heights that do not divide evenly.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from volume_renderer_amd.scheduler import RowScheduler, assemble, rows_of_rank, rows_per_rank_max  # noqa: E402


def test_row_partition_covers_every_row_once():
    for height in (1, 7, 54, 1080, 2160):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                first, stride, count = rows_of_rank(r, world, height)
                seen += [first + k * stride for k in range(count)]
                assert count <= rows_per_rank_max(world, height)
            assert sorted(seen) == list(range(height))
    with pytest.raises(ValueError):
        rows_of_rank(2, 2, 10)


def test_assemble_inverts_the_interleave():
    h, w, world = 11, 5, 4
    frame = torch.arange(h * w * 3, dtype=torch.float32).reshape(h, w, 3)
    rows_max = rows_per_rank_max(world, h)
    g = torch.zeros((world, rows_max, w, 3))
    for r in range(world):
        part = frame[r::world]
        g[r, :part.shape[0]] = part
    assert torch.equal(assemble(g, h, world), frame)


def _worker(rank, world, port, height, width, out_path):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle_lib as ol
    import scenes
    from volume_renderer_amd.abi import Grid

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        L = ol.lib()
        n = 32
        vox = np.zeros((n, n, n), dtype=np.float32)
        L.vro_make_cloud(n, scenes.cloud_radius(n), ol.fptr(vox), 2)
        scene, _keep = ol.make_scene(density=Grid(vox))
        opts = scenes.options_cloud("ratioTracking", n, width, height, 4)
        sched = RowScheduler(height, width, torch.device("cpu"))
        assert (sched.rank, sched.world) == (rank, world)

        def render_rows(first, stride, count, out):
            for k in range(count):
                j = first + k * stride
                pix, _, _ = ol.render_rows(scene, opts, 2, j, j + 1, rng_mode=ol.RNG_PHILOX, seed=scenes.SEED)
                out[k] = torch.from_numpy(pix[0])

        frame = sched.render(render_rows)
        gmax = sched.global_max(float(sched.part.max()))
        if rank == 0:
            np.save(out_path, np.concatenate([frame.numpy().ravel(), [gmax]]).astype(np.float32))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,height", [(2, 14), (3, 10)])
def test_gloo_ranks_assemble_the_single_process_frame(tmp_path, world, height):
    import oracle_lib as ol
    import scenes
    from volume_renderer_amd.abi import Grid

    width = 12
    out_path = str(tmp_path / "frame.npy")
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, height, width, out_path), nprocs=world, join=True)
    got = np.load(out_path)
    L = ol.lib()
    n = 32
    vox = np.zeros((n, n, n), dtype=np.float32)
    L.vro_make_cloud(n, scenes.cloud_radius(n), ol.fptr(vox), 2)
    scene, _keep = ol.make_scene(density=Grid(vox))
    opts = scenes.options_cloud("ratioTracking", n, width, height, 4)
    want = ol.render(scene, opts, 2, rng_mode=ol.RNG_PHILOX, seed=scenes.SEED)
    assert np.array_equal(got[:-1].reshape(height, width, 3).view(np.uint32), want.view(np.uint32))
    assert got[-1] == want.max()           # tonemap white point = max over all ranks' rows
