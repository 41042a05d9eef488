#!/usr/bin/env python3
"""bench.py — path-samples/s of the integrator hot path on MI355X (BASELINE.json metric).

A "step" is one full frame of the workload through the HIP path: every rank renders its
interleaved row set (row r, r+N, r+2N, ... of the frame) into HBM, then the framebuffer is
gathered on rank 0 with one RCCL all_gather.  Default workload = BASELINE config 2: ratio
tracking, 1920x1080 @ 64 spp, 512^3 procedural cloud, the step parameters of the reference's
bunny-cloud-configRenderer.xml.  The density grid is generated on the device before the
timed region, so inputs are resident in HBM when timing starts.

Scaling over N GPUs (--scaling):
  weak   (default) the frame keeps its resolution and every GPU keeps its sample count: the frame is rendered
         at spp x N samples per pixel, rank r taking rows r::N — N GPUs buy N times the samples in the same
         time.  At N = 1 this IS the named configuration.
  strong the named frame itself (spp as configured) split by rows over the ranks; bounded by the fixed tail of
         the longest paths.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task description) with two extra objects:
  roofline      algorithmic density-lookup bytes (4 B per lookup, counted in-kernel) over the
                HIP-event time of the integrator kernel, against the 8 TB/s HBM peak
  cpu_baseline  the CPU oracle (a port of the reference algorithm) on the host cores, on a
                bounded sample of the same frame (N=1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402  (first: the backend shares torch's HIP runtime)
import torch.distributed as dist  # noqa: E402

HBM_PEAK_BYTES_PER_S = 8.0e12  # MI355X_MICROARCH.md: HBM3E 8 TB/s
BYTES_PER_LOOKUP = 4           # one f32 voxel per density lookup

WORKLOADS = {
    # name: (integrator, grid n, width, height, spp, regime)
    "config2_ratio_1080p_64spp_cloud512": ("ratioTracking", 512, 1920, 1080, 64, "shipped"),
    "config2_textbook": ("ratioTracking", 512, 1920, 1080, 64, "textbook"),
    "config3_delta_1080p_256spp_cloud512": ("deltaTracking", 512, 1920, 1080, 256, "shipped"),
    "config3_textbook": ("deltaTracking", 512, 1920, 1080, 256, "textbook"),
    "config4_delta_4k_256spp_cloud1024": ("deltaTracking", 1024, 3840, 2160, 256, "textbook"),
    "small": ("ratioTracking", 128, 480, 270, 4, "shipped"),
    # fixed-step grid marcher on the same cloud: one lookup per half-voxel step + light marches
    "gridmarch_simple_1080p_cloud512": ("nanoVDBSimple", 512, 1920, 1080, 4, "shipped"),
    # mesh-bounded homogeneous media (BASELINE configs 1 and 5): (integrator, mesh, width, height, spp, -)
    "config1_marcher_cube_512_1spp": ("homogeneousRayMarcherNEE", "cube", 512, 512, 1, "mesh"),
    "config5_marcher_sphere_4k": ("homogeneousRayMarcherImproved", "icosphere", 3840, 2160, 64, "mesh"),
    "config5_marcher_nee_sphere_4k": ("homogeneousRayMarcherNEE", "icosphere", 3840, 2160, 64, "mesh"),
    "perlin_marcher_sphere_1080p": ("heterogeneousPerlinNoise", "icosphere", 1920, 1080, 16, "mesh"),
}
BVH_NODE_BYTES = 36            # one BvhNode per closest-hit query at the least (vr_kernels.h)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="config2_ratio_1080p_64spp_cloud512", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="fast", choices=["fast", "exact"])
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"], help="see the module docstring")
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "on", "off"])
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target CPU work for the baseline sample")
    return ap.parse_args()


def cpu_baseline(integ_id, n, opts, target_seconds, mesh=None):
    """Oracle (port of the reference algorithm) on the host cores, bounded sample of the same frame."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor

    sys.path.insert(0, os.path.join(ROOT, "tests"))      # the oracle's loader: this leg only
    import oracle_lib as ol
    from volume_renderer_amd.abi import Grid, copy_options
    from volume_renderer_amd import workloads as scenes

    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16") or 16), 64)
    L = ol.lib()
    L.vro_set_math_mode(ol.MATH_LIBM)
    if mesh is not None:
        scene, _keep = ol.make_scene(mesh=mesh)
    else:
        vox = np.zeros((n, n, n), dtype=np.float32)
        L.vro_make_cloud(n, scenes.cloud_radius(n), ol.fptr(vox), cores)
        grid = Grid(vox)
        scene, _keep = ol.make_scene(density=grid)
    o1 = copy_options(opts, rayPerPixelCount=1)   # 1 sub-sample per pixel, scaled linearly in spp

    def rows(row_list):
        def one(j):
            t = time.perf_counter()
            ol.render_rows(scene, o1, integ_id, j, j + 1, rng_mode=ol.RNG_REFERENCE)
            return time.perf_counter() - t
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            list(ex.map(one, row_list))
        return time.perf_counter() - t0

    H, W = opts.height, opts.width
    # calibrate on a sparse row set, then size the sample to ~target_seconds
    probe = list(range(H // 128, H, max(1, H // 32)))
    tp = rows(probe)
    per_row = tp / len(probe)
    nrows = int(max(len(probe), min(H, target_seconds / max(per_row, 1e-6))))
    stride = max(1, H // nrows)
    sample = list(range(stride // 2, H, stride))
    ts = rows(sample)
    samples = len(sample) * W
    return {
        "value": samples / ts / 1e6,
        "unit": "Mpath-samples/s",
        "cores": cores,
        "kind": "port",
        "sample": f"rows {stride // 2}::{stride} of the {W}x{H} frame ({len(sample)} rows), 1 of {opts.rayPerPixelCount} "
                  f"sub-samples per pixel, oracle/vr_oracle.c (libm, mt19937 per row) on {cores} threads, {ts:.1f} s",
    }


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    # VR_BENCH_REHEARSAL=1: every rank on GPU 0 and the exchange over gloo on host tensors — a way to walk the
    # N > 1 path on a one-GPU box (RCCL refuses two ranks on one device); never a measurement
    rehearsal = os.environ.get("VR_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from volume_renderer_amd import MODE_EXACT, MODE_FAST, Backend, INTEGRATORS
    from volume_renderer_amd import workloads as scenes

    integ, n, W, H, spp, regime = WORKLOADS[a.workload]
    if a.spp:
        spp = a.spp
    spp_config = spp
    if a.scaling == "weak":
        spp *= world                                 # per-GPU sample count fixed: rows r::N at spp x N
    integ_id = INTEGRATORS[integ]
    mesh_scene = regime == "mesh"

    be = Backend(local)
    be.set_mode(MODE_EXACT if a.mode == "exact" else MODE_FAST)
    be.set_seed(scenes.SEED)
    if mesh_scene:
        opts = scenes.options_marcher(W, H, spp)
        mesh = scenes.cube_mesh() if n == "cube" else scenes.icosphere_mesh()
        be.set_mesh(mesh)                            # BVH resident in HBM before timing
    else:
        opts = scenes.options_cloud(integ, n, W, H, spp, regime=regime)
        if os.environ.get("VR_BENCH_CAMERA_DIST"):   # developer knob: camera distance in cloud radii (default 2.2)
            opts.cameraPosition[2] = float(os.environ["VR_BENCH_CAMERA_DIST"]) * scenes.cloud_radius(n)
        be.generate_cloud(n, scenes.cloud_radius(n))     # density resident in HBM before timing

    from volume_renderer_amd.scheduler import RowScheduler
    sched = RowScheduler(H, W, torch.device("cpu") if rehearsal else dev, rank=rank, world=world)
    staging = torch.zeros_like(sched.part, device=dev) if rehearsal else None

    def render_rows(first, stride, count, out):
        target = staging if rehearsal else out
        be.render_row_set_device(opts, integ_id, first, stride, count, target.data_ptr())
        be.synchronize()
        if rehearsal:
            out.copy_(staging)

    def step():
        return sched.render(render_rows)

    frame = sched.frame

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    kernel_ms = 0.0
    launches = 0
    lookups = 0
    stats = None
    for _ in range(a.steps):
        step()
        stats = be.last_stats()
        kernel_ms += stats["kernel_ms"]
        launches += stats["kernel_launches"]
        lookups += stats["ray_queries"] if mesh_scene else stats["density_lookups"]
    fence()
    dt = time.perf_counter() - t0
    cdev = torch.device("cpu") if rehearsal else dev
    tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
    agg = torch.tensor([float(lookups), float(kernel_ms), float(launches)], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
    dt = float(tt.item())
    lookups_all, kernel_ms_all, launches_all = (float(x) for x in agg.tolist())

    if rank == 0:
        samples_per_step = W * H * spp
        value = samples_per_step * a.steps / dt / 1e6
        # dominant kernel = the tracking kernel; per-launch averages over every rank's launches
        avg_launch_s = kernel_ms_all / 1e3 / max(launches_all, 1.0)
        bytes_per_launch = lookups_all * (BVH_NODE_BYTES if mesh_scene else BYTES_PER_LOOKUP) / max(launches_all, 1.0)
        achieved = bytes_per_launch / max(avg_launch_s, 1e-12) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            with open(pmc) as f:
                traffic = json.load(f).get(a.workload, {}).get("hbm_bytes_per_launch")
        out = {
            "metric": "Mpath-samples/sec at 1080p (1/2/4/8 GPU) + density lookups/s vs HBM roofline",
            "value": value,
            "unit": "Mpath-samples/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": a.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": a.workload, "integrator": integ, "image": f"{W}x{H}", "spp": spp, "spp_at_1_gpu": spp_config,
                "grid": (f"{n} mesh, threaded BVH" if mesh_scene else
                         f"{n}^3 procedural cloud (Perlin fbm), bricked 8^3, replicated per GPU"), "step_regime": regime,
                "kernel_mode": a.mode,
                "partition": ("REHEARSAL: all ranks on one GPU, gloo" if rehearsal else f"rows r::{world} per rank, RCCL all_gather") if world > 1 else "single GPU",
                "frame_mean_rgb": [float(x) for x in frame.mean(dim=(0, 1)).tolist()],
            },
            "glookups_per_s": lookups_all / max(kernel_ms_all / 1e3 / world, 1e-12) / 1e9,
            "lookups_per_sample": lookups_all / (samples_per_step * a.steps),
            "divergence": {k: stats[k] for k in ("wave_trips", "wave_trip_lanes", "event_phases", "scatter_events")}
            | {"walking_lanes_per_trip": stats["wave_trip_lanes"] / max(stats["wave_trips"], 1)},
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_BYTES_PER_S / 1e9, "unit": "GB/s",
                "frac": achieved * 1e9 / HBM_PEAK_BYTES_PER_S, "traffic": traffic,
                "kernel": {0: "vr_march_kernel", 1: "vr_march_kernel", 6: "vr_march_kernel", 4: "vr_gridmarch_kernel", 5: "vr_gridmarch_kernel"}.get(integ_id, "vr_track_kernel"),
                "kernel_ms_per_launch": avg_launch_s * 1e3, "launches": int(launches_all),
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "note": ("algorithmic bytes = 36 B (one BVH node) x closest-hit queries: a lower bound, the BVH is L2-resident"
                         if mesh_scene else
                         "algorithmic bytes = 4 B x in-kernel lookup count; sub-voxel steps are served from a register, so "
                         "frac measures lookup rate, not HBM traffic (see traffic)"),
            },
        }
        want_cpu = a.cpu_baseline == "on" or (a.cpu_baseline == "auto" and world == 1)
        if want_cpu:
            try:
                out["cpu_baseline"] = cpu_baseline(integ_id, n, opts, a.cpu_seconds, mesh if mesh_scene else None)
            except Exception as e:  # the baseline is reported, never required for the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "Mpath-samples/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
