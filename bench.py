#!/usr/bin/env python3
"""bench.py — path-samples/s of the integrator hot path on MI355X (BASELINE.json metric).

A "step" is one full frame of the workload through the HIP path: every rank renders its
interleaved row set (row r, r+N, r+2N, ... of the frame) into HBM, then the framebuffer is
gathered on rank 0 with one RCCL all_gather.  Default workload = BASELINE config 2: ratio
tracking, 1920x1080 @ 64 spp, 512^3 procedural cloud, the step parameters of the reference's
bunny-cloud-configRenderer.xml.  The density grid is generated on the device before the
timed region, so inputs are resident in HBM when timing starts.

Scaling over N GPUs (--scaling):
  strong (default) the NAMED frame itself (1080p at 64 spp for config 2) split by rows over the ranks, rank r
         taking rows r::N; bounded by the fixed tail of the longest paths.  With N > 1 the line
         carries an extra object "weak_scaling" measured right after the timed steps: the same frame at
         spp x N samples per pixel (every GPU keeps its sample count).
  weak   only the weak measurement, as the line's value.  At N = 1 both ARE the named configuration.
The timed step ends with the gathered frame copied to host memory on rank 0 ("to gathered framebuffer
on host").  The line's value is measured one frame at a time (a frame is complete before the next one starts, as a call
of the reference's render() is).  The extra object "two_frames_in_flight" is the same K frames with the next frame's
kernel queued on a second backend (stream) while the current one drains: a launch ends with a tail of a few very long
paths, and the next frame's waves fill it.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task description) with two extra objects:
  roofline      of the integrator kernel, timed with HIP events on the backend's stream.  `bound` names what bounds it and
                achieved / peak / frac are OF THAT BOUND, never above 1: "hbm" — algorithmic bytes (4 B per density lookup,
                counted in-kernel) against 8 TB/s, where every lookup is a load; "valu" — VALU instructions per second
                against the issue peak (1024 SIMDs x clock / 2), where sub-voxel steps serve most lookups from a register;
                "hbm-latency" — HBM bytes (PMC) against 8 TB/s, where waves wait for random 4-byte gathers most of their
                cycles.  Inside: `lookup_rate` = SURVEY 8(d)'s algorithmic figure (a rate), `loads` = 4 B x the lookups
                that issued a load, `hbm` = traffic from the committed PMC passes (= `traffic`), `valu` = instruction rate,
                lane utilisation, scalar instructions, share of wave cycles waiting (profiles/pmc_traffic.json)
  cpu_baseline  the CPU oracle (a port of the reference algorithm) on the host cores, on a
                bounded sample of the same frame (N=1 only).  Its `cross_check` compares the sample's rows (one sub-sample
                per pixel, the reference's own generator) with the same rows of the frame the timed steps rendered: image
                means, relative difference and z = mean difference over its standard error across pixels

--client group (single process, no torch.distributed): the frame over GPUs 0..N-1 through vr_group_render, the client
behind the C-ABI that INTEGRATION.md 6 recommends (BaseIntegrator.cpp:143-181 replaced in one call); VR_BENCH_GROUP_DEVICES
"0,0,0,0" lists the devices explicitly (a rehearsal on one GPU: never a measurement of scaling).
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
# MI355X_MICROARCH.md "Wave scheduling": 256 CUs x 4 SIMDs, a wave issues a VALU instruction over 2 cycles; 2.4 GHz
VALU_PEAK_WAVE_INSTS_PER_S = 256 * 4 * 2.4e9 / 2
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
    # config 5's other asset: the reference's Models/CornellBox-Sphere.obj (2 192 triangles; arrays in tests/golden/ref_models.npz)
    "config5_marcher_cornell_4k": ("homogeneousRayMarcherImproved", "cornell_sphere", 3840, 2160, 64, "mesh"),
    "config5_marcher_nee_cornell_4k": ("homogeneousRayMarcherNEE", "cornell_sphere", 3840, 2160, 64, "mesh"),
    # config 5, tracker leg: the trackers bounded by the mesh (no reference counterpart): the n^3 cloud laid
    # over the 320-triangle sphere, seen by the marchers' camera
    "config5_tracker_ratio_sphere_4k": ("meshRatioTracking", 1024, 3840, 2160, 16, "shipped"),
    "config5_tracker_delta_sphere_4k": ("meshDeltaTracking", 1024, 3840, 2160, 256, "textbook"),
    "config5_tracker_delta_shipped_sphere_4k": ("meshDeltaTracking", 1024, 3840, 2160, 64, "shipped"),
}
BVH_NODE_BYTES = 32            # one BvhNode per closest-hit query at the least (vr_kernels.h)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="config2_ratio_1080p_64spp_cloud512", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="fast", choices=["fast", "exact", "local"],
                    help="local = VR_MODE_FAST_LOCAL: delta tracking over per-brick majorants (textbook regimes only; not the reference's estimator)")
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"], help="see the module docstring")
    ap.add_argument("--pipeline", default="auto", choices=["auto", "on", "off"],
                    help="also measure the K frames with two in flight (extra object; auto: tracking workloads, not in rehearsal)")
    ap.add_argument("--client", default="ranks", choices=["ranks", "group"],
                    help="ranks: one process per GPU, RCCL all_gather (the driver's launch); group: ONE process, vr_group_render over --gpus devices")
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "on", "off"])
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target CPU work for the baseline sample")
    return ap.parse_args()


def cpu_baseline(integ_id, n, opts, target_seconds, mesh=None, gpu_frame=None, local=False):
    """Oracle (port of the reference algorithm) on the host cores, bounded sample of the same frame.  With the GPU's frame at
    hand the sample doubles as a full-size cross-check: the oracle's rows (one sub-sample per pixel, the reference's own
    generator) against the same rows of the frame the timed steps rendered.  The local-majorant modes are not the reference's estimator
    (+1 % image mean on the 512^3 cloud); their frames are checked against the oracle's restatement of THAT tracker,
    rendered over the same rows after the timed sample (the baseline's value stays the reference algorithm's)."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor

    sys.path.insert(0, os.path.join(ROOT, "tests"))      # the oracle's loader: this leg only
    import oracle_lib as ol
    from volume_renderer_amd.abi import Grid, copy_options
    from volume_renderer_amd import workloads as scenes

    # every core this process may run on; the GPU box hands a one-GPU job a 16-core share of a much larger host
    # (OMP_NUM_THREADS there, which worker pools must respect), so that is the honest "host cores" of the run
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    cores = max(1, min(avail, share) if share > 0 else avail)
    L = ol.lib()
    L.vro_set_math_mode(ol.MATH_LIBM)
    if mesh is not None and not isinstance(n, int):
        scene, _keep = ol.make_scene(mesh=mesh)
    else:
        vox = np.zeros((n, n, n), dtype=np.float32)
        L.vro_make_cloud(n, scenes.cloud_radius(n), ol.fptr(vox), cores)
        if mesh is not None:      # mesh-bounded trackers: the cloud laid over the mesh
            vs, origin = scenes.mesh_tracker_grid_map(n)
            grid = Grid(vox, voxel_size=vs, origin=origin)
        else:
            grid = Grid(vox)
        scene, _keep = ol.make_scene(density=grid, mesh=mesh)
    o1 = copy_options(opts, rayPerPixelCount=1)   # 1 sub-sample per pixel, scaled linearly in spp

    got = {}

    def rows(row_list):
        def one(j):
            t = time.perf_counter()
            got[j] = ol.render_rows(scene, o1, integ_id, j, j + 1, rng_mode=ol.RNG_REFERENCE)[0][0]
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
    check = None
    if gpu_frame is not None:
        which = "reference scheme"
        if local:
            ol.local_majorants(scene)
            local_id = ol.MESH_DELTA_LOCAL if (mesh is not None and isinstance(n, int)) else ol.DELTA_LOCAL
            which = "local-majorant tracker, reference generator"

            def one_local(j):
                got[j] = ol.render_rows(scene, o1, local_id, j, j + 1, rng_mode=ol.RNG_REFERENCE)[0][0]
            with ThreadPoolExecutor(cores) as ex:
                list(ex.map(one_local, sorted(got)))
            L.vro_local_majorants_free()
        js = sorted(got)
        cpu = np.stack([got[j] for j in js]).astype(np.float64)              # (rows, W, 3), 1 sub-sample per pixel
        gpu = np.asarray(gpu_frame, dtype=np.float64)[js]                    # the same rows at the frame's sample count
        d = (cpu - gpu).mean(axis=2).ravel()
        se = d.std(ddof=1) / np.sqrt(d.size)
        check = {"rows": len(js), "oracle_mean": float(cpu.mean()), "gpu_mean": float(gpu.mean()),
                 "rel_diff": float((cpu.mean() - gpu.mean()) / max(abs(gpu.mean()), 1e-30)), "z": float(d.mean() / max(se, 1e-30)),
                 "what": f"mean radiance over the sampled rows: oracle ({which}, 1 sub-sample per pixel) against the frame the "
                         "timed steps rendered; z = mean difference over its standard error across pixels"}
    return {
        "cross_check": check,
        "value": samples / ts / 1e6,
        "unit": "Mpath-samples/s",
        "cores": cores,
        "host_cpus": os.cpu_count(),
        "kind": "port",
        "sample": f"rows {stride // 2}::{stride} of the {W}x{H} frame ({len(sample)} rows), 1 of {opts.rayPerPixelCount} "
                  f"sub-samples per pixel, oracle/vr_oracle.c (libm, mt19937 per row) on {cores} threads, {ts:.1f} s",
    }


def main_group(a):
    """--client group: K frames through vr_group_render, each complete in (pageable) host memory when the call returns."""
    from volume_renderer_amd import MODE_EXACT, MODE_FAST, MODE_FAST_LOCAL, BackendGroup, INTEGRATORS
    from volume_renderer_amd import workloads as scenes
    integ, n, W, H, spp, regime = WORKLOADS[a.workload]
    if a.spp:
        spp = a.spp
    integ_id = INTEGRATORS[integ]
    devs = [int(x) for x in os.environ.get("VR_BENCH_GROUP_DEVICES", ",".join(str(i) for i in range(a.gpus))).split(",")]
    rehearsal = len(set(devs)) != len(devs)
    g = BackendGroup(devs)
    g.set_mode({"exact": MODE_EXACT, "local": MODE_FAST_LOCAL}.get(a.mode, MODE_FAST))
    g.set_seed(scenes.SEED)
    if regime == "mesh":
        opts = scenes.options_marcher(W, H, spp)
        if n == "cornell_sphere":
            mesh = scenes.reference_model(n, os.path.join(ROOT, "tests", "golden", "ref_models.npz"))
            opts.cameraPosition[:] = (0.0, 0.8, 3.6)
            opts.fov = 40.0
        else:
            mesh = scenes.cube_mesh() if n == "cube" else scenes.icosphere_mesh()
        g.set_mesh(mesh)
    else:
        g.generate_cloud(n, scenes.cloud_radius(n))
        if integ.startswith("mesh"):
            opts = scenes.options_mesh_tracker(integ, n, W, H, spp, regime=regime)
            g.set_density_map(*scenes.mesh_tracker_grid_map(n))
            g.set_mesh(scenes.icosphere_mesh())
        else:
            opts = scenes.options_cloud(integ, n, W, H, spp, regime=regime)
    frame = np.zeros((H, W, 3), dtype=np.float32)            # pageable, like the reference's `new Vec3f[width * height]`
    for _ in range(a.warmup):
        g.render(opts, integ_id, out=frame)
    t0 = time.perf_counter()
    kernel_ms = 0.0
    for _ in range(a.steps):
        g.render(opts, integ_id, out=frame)
        kernel_ms += g.last_stats()["kernel_ms"]
    dt = time.perf_counter() - t0
    info = g.gather_info()
    out = {"metric": "Mpath-samples/sec at 1080p (1/2/4/8 GPU) + density lookups/s vs HBM roofline",
           "value": W * H * spp * a.steps / dt / 1e6, "unit": "Mpath-samples/s", "n_gpus": len(devs), "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic",
           "config": {"workload": a.workload, "integrator": integ, "image": f"{W}x{H}", "spp": spp, "kernel_mode": a.mode,
                      "partition": ("REHEARSAL: a GPU listed several times; " if rehearsal else "") +
                                   f"one process, vr_group_render over devices {devs}, rows i::{len(devs)}, gather by 2-D copies into the caller's frame",
                      "timed_span": "vr_group_render: render on every device + gather into host memory",
                      "frame_mean_rgb": [float(x) for x in frame.mean(axis=(0, 1))]},
           "group": {"slowest_device_kernel_ms_per_step": kernel_ms / a.steps, "gather_and_launch_ms_per_step": (dt * 1e3 - kernel_ms) / a.steps, **info}}
    print(json.dumps(out), flush=True)
    g.close()


def main():
    a = parse()
    if a.client == "group":
        if int(os.environ.get("WORLD_SIZE", "1")) != 1:
            raise SystemExit("--client group runs in ONE process (no torch.distributed launcher)")
        return main_group(a)
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

    from volume_renderer_amd import MODE_EXACT, MODE_FAST, MODE_FAST_LOCAL, Backend, INTEGRATORS
    from volume_renderer_amd import workloads as scenes
    from volume_renderer_amd.abi import copy_options

    integ, n, W, H, spp, regime = WORKLOADS[a.workload]
    if a.spp:
        spp = a.spp
    integ_id = INTEGRATORS[integ]
    mesh_scene = regime == "mesh"
    mesh_tracker = integ.startswith("mesh")

    be = Backend(local)
    be.set_mode({"exact": MODE_EXACT, "local": MODE_FAST_LOCAL}.get(a.mode, MODE_FAST))
    be.set_seed(scenes.SEED)
    mesh = None
    if mesh_scene:
        opts = scenes.options_marcher(W, H, spp)
        if n == "cornell_sphere":
            mesh = scenes.reference_model(n, os.path.join(ROOT, "tests", "golden", "ref_models.npz"))   # data, not code
            opts.cameraPosition[:] = (0.0, 0.8, 3.6)     # the box is [-1.02, 1] x [0, 1.59] x [-1.04, 0.99]: fill the frame with it
            opts.fov = 40.0
        else:
            mesh = scenes.cube_mesh() if n == "cube" else scenes.icosphere_mesh()
        be.set_mesh(mesh)                            # BVH resident in HBM before timing
    else:
        be.generate_cloud(n, scenes.cloud_radius(n))     # density resident in HBM before timing
        if mesh_tracker:
            opts = scenes.options_mesh_tracker(integ, n, W, H, spp, regime=regime)
            mesh = scenes.icosphere_mesh()
            be.set_density_map(*scenes.mesh_tracker_grid_map(n))
            be.set_mesh(mesh)
        else:
            opts = scenes.options_cloud(integ, n, W, H, spp, regime=regime)
            if os.environ.get("VR_BENCH_CAMERA_DIST"):   # developer knob: camera distance in cloud radii (default 2.2)
                opts.cameraPosition[2] = float(os.environ["VR_BENCH_CAMERA_DIST"]) * scenes.cloud_radius(n)

    from volume_renderer_amd.scheduler import RowScheduler
    sched = RowScheduler(H, W, torch.device("cpu") if rehearsal else dev, rank=rank, world=world)
    staging = torch.zeros_like(sched.part, device=dev) if rehearsal else None
    # the frame's place in host memory (rank 0): the timed step ends there, as BaseIntegrator::render's span does
    host_frame = torch.empty((H, W, 3), dtype=torch.float32, pin_memory=True) if rank == 0 else None
    cdev = torch.device("cpu") if rehearsal else dev

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run(o, steps, warmup):
        """`warmup` untimed and `steps` timed frames of options `o`; max-over-ranks time and per-launch sums."""
        def render_rows(first, stride, count, out):
            target = staging if rehearsal else out
            be.render_row_set_device(o, integ_id, first, stride, count, target.data_ptr())
            be.synchronize()
            if rehearsal:
                out.copy_(staging)

        def step():
            frame = sched.render(render_rows)
            if rank == 0:
                host_frame.copy_(frame)              # device -> pinned host, synchronous
            # every rank: the gather reads `part` on torch's stream, the next frame's kernel writes it on the backend's own
            # (non-blocking) stream — wait for the gather before launching again (include/vr_backend.h, render_row_set_device)
            torch.cuda.current_stream().synchronize()
            return frame

        for _ in range(warmup):
            step()
        fence()
        t0 = time.perf_counter()
        kernel_ms = 0.0
        launches = 0
        lookups = 0
        loads = 0
        stats = None
        for _ in range(steps):
            step()
            stats = be.last_stats()
            kernel_ms += stats["kernel_ms"]
            launches += stats["kernel_launches"]
            lookups += stats["ray_queries"] if mesh_scene else stats["density_lookups"]
            loads += stats["voxel_loads"]
        fence()
        dt = time.perf_counter() - t0
        tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
        agg = torch.tensor([float(lookups), float(kernel_ms), float(launches), float(loads)], dtype=torch.float64, device=cdev)
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        r = dict(zip(("lookups", "kernel_ms", "launches", "loads"), (float(x) for x in agg.tolist())))
        r.update(dt=float(tt.item()), stats=stats, samples_per_step=W * H * o.rayPerPixelCount)
        return r

    weak = None
    if a.scaling == "weak" and world > 1:
        opts = copy_options(opts, rayPerPixelCount=spp * world)      # per-GPU sample count fixed: rows r::N at spp x N
        main = run(opts, a.steps, a.warmup)
        frame_mean = [float(x) for x in sched.frame.mean(dim=(0, 1)).tolist()] if rank == 0 else None
    else:
        main = run(opts, a.steps, a.warmup)                           # the named frame, split by rows
        frame_mean = [float(x) for x in sched.frame.mean(dim=(0, 1)).tolist()] if rank == 0 else None
        if world > 1:
            w = run(copy_options(opts, rayPerPixelCount=spp * world), a.steps, 1)
            weak = {"value": w["samples_per_step"] * a.steps / w["dt"] / 1e6, "unit": "Mpath-samples/s", "spp": spp * world,
                    "ms_per_step": w["dt"] / a.steps * 1e3,
                    "note": "same frame at spp x N samples per pixel, rows r::N per rank; measured after the timed steps"}
    # the same K frames, two in flight: frame k renders on backend k % 2 while frame k - 1 is gathered and copied out
    piped = None
    # (auto: single-GPU runs only — at N > 1 the extra leg would put a second set of collectives between the timed steps and
    # the JSON line; `--pipeline on` asks for it explicitly)
    want_pipe = a.pipeline == "on" or (a.pipeline == "auto" and world == 1 and not rehearsal and not mesh_scene)
    if want_pipe:
        be2 = Backend(local)
        be2.set_mode({"exact": MODE_EXACT, "local": MODE_FAST_LOCAL}.get(a.mode, MODE_FAST))
        be2.set_seed(scenes.SEED)
        if mesh_scene:
            be2.set_mesh(mesh)
        else:
            be2.generate_cloud(n, scenes.cloud_radius(n))
            if mesh_tracker:
                be2.set_density_map(*scenes.mesh_tracker_grid_map(n))
                be2.set_mesh(mesh)
        bes = [be, be2]
        scheds = [sched, RowScheduler(H, W, dev, rank=rank, world=world)]
        from volume_renderer_amd.scheduler import rows_of_rank
        first, stride, count = rows_of_rank(rank, world, H)

        def launch(k):
            bes[k % 2].render_row_set_device(opts, integ_id, first, stride, count, scheds[k % 2].part.data_ptr())

        def finish(k):
            bes[k % 2].synchronize()
            frame = scheds[k % 2].gather()
            if rank == 0:
                host_frame.copy_(frame)
            torch.cuda.current_stream().synchronize()   # the gather has read `part` before launch(k + 2) writes it

        for k in range(2):                      # warm-up: both backends' buffers exist
            launch(k); finish(k)
        fence()
        t0 = time.perf_counter()
        for k in range(a.steps):
            launch(k)
            if k > 0:
                finish(k - 1)
        finish(a.steps - 1)
        fence()
        tp = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=cdev)
        if world > 1:
            dist.all_reduce(tp, op=dist.ReduceOp.MAX)
        dtp = float(tp.item())
        piped = {"value": W * H * int(opts.rayPerPixelCount) * a.steps / dtp / 1e6, "unit": "Mpath-samples/s", "ms_per_step": dtp / a.steps * 1e3,
                 "steps": a.steps, "note": "the same K frames of the named workload, frame k + 1 queued on a second backend while frame k "
                 "drains its tail and is gathered; every frame complete and on the host inside the timed region"}
        be2.close()

    if rank == 0:
        dt, stats = main["dt"], main["stats"]
        lookups_all, kernel_ms_all, launches_all, loads_all = main["lookups"], main["kernel_ms"], main["launches"], main["loads"]
        samples_per_step = main["samples_per_step"]
        value = samples_per_step * a.steps / dt / 1e6
        # dominant kernel = the integrator kernel; per-launch averages over every rank's launches
        nl = max(launches_all, 1.0)
        avg_launch_s = max(kernel_ms_all / 1e3 / nl, 1e-12)
        bytes_per_launch = lookups_all * (BVH_NODE_BYTES if mesh_scene else BYTES_PER_LOOKUP) / nl
        achieved = bytes_per_launch / avg_launch_s / 1e9
        pmc = {}
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc_path):
            with open(pmc_path) as f:
                pmc = json.load(f).get(a.workload if a.mode == "fast" else f"{a.workload}:{a.mode}", {})
        use_pmc = world == 1 and not a.spp and bool(pmc)             # the committed counters are per launch of THIS frame in THIS mode
        traffic = pmc.get("hbm_bytes_per_launch") if use_pmc else None
        kernel_name = {0: "vr_march_kernel", 1: "vr_march_kernel", 6: "vr_march_kernel", 7: "vr_march_kernel", 4: "vr_gridmarch_kernel",
                       5: "vr_gridmarch_kernel"}.get(integ_id, "vr_track_kernel")
        if a.mode == "local" and integ in ("deltaTracking", "meshDeltaTracking"):
            kernel_name = "vr_delta_dda_kernel"
        peak_gbs = HBM_PEAK_BYTES_PER_S / 1e9
        # SURVEY 8(d)'s figure — 4 B x lookups as the reference makes them (32 B x closest-hit queries for the mesh marchers) over
        # the kernel time — is a lookup RATE wherever the kernel serves lookups without memory (a step that stays in its voxel reads
        # a register, empty-space and closed-form runs are arithmetic): kept as `lookup_rate`, never as a fraction of a roofline.
        lookup_rate = {"bytes_per_launch": bytes_per_launch, "GBps": achieved, "x_hbm_peak": achieved / peak_gbs,
                       "what": ("32 B (one BVH node) x closest-hit queries: a lower bound on what a query reads, and the BVH is cache-resident"
                                if mesh_scene else "4 B x density lookups counted as the reference makes them (one per step of every walk)")}
        loads_obj = None
        loads_per_lookup = 1.0
        if not mesh_scene:
            loaded = loads_all * BYTES_PER_LOOKUP / nl
            loads_per_lookup = loads_all / max(lookups_all, 1.0)
            loads_obj = {"achieved": loaded / avg_launch_s / 1e9, "peak": peak_gbs, "unit": "GB/s", "frac": loaded / avg_launch_s / HBM_PEAK_BYTES_PER_S,
                         "bytes_per_launch": loaded, "loads_per_lookup": loads_per_lookup,
                         "what": "4 B x the lookups that issued a load (counted in the kernel): what the memory system is asked for"}
        hbm_obj = None
        if traffic:
            hbm_obj = {"achieved": traffic / avg_launch_s / 1e9, "peak": peak_gbs, "unit": "GB/s", "frac": traffic / avg_launch_s / HBM_PEAK_BYTES_PER_S,
                       "bytes_per_launch": traffic, "source": pmc.get("source"),
                       "what": "HBM bytes per launch from the committed PMC passes (FETCH_SIZE x 2 + WRITE_SIZE, in KiB) over the live kernel time"}
        valu_obj = None
        if use_pmc and pmc.get("valu_insts_per_launch"):
            rate = pmc["valu_insts_per_launch"] / avg_launch_s
            valu_obj = {"achieved": rate / 1e9, "peak": VALU_PEAK_WAVE_INSTS_PER_S / 1e9, "unit": "G wave-instr/s",
                        "frac": rate / VALU_PEAK_WAVE_INSTS_PER_S, "insts_per_launch": pmc["valu_insts_per_launch"],
                        "source": pmc.get("valu_source"),
                        "note": "peak = one VALU instruction per 2 cycles per SIMD (nominal); compares, conversions, min/max and three-input bit ops "
                                "issue at half that rate, log/exp/rcp at a quarter (profiles/r02_ubench_valu.txt), and the CU's one scalar unit "
                                "takes issue slots beside them (profiles/r03_ubench_salu.txt): a tracking kernel's mix saturates near frac 0.55"}
            for k in ("valu_lane_utilisation", "salu_insts_per_launch", "wave_cycles_waiting"):
                if pmc.get(k) is not None:
                    valu_obj[k.replace("valu_", "")] = pmc[k]
        # What bounds the kernel, and the fraction OF THAT BOUND (never above 1):
        #  - "hbm": the contract's form, algorithmic bytes against the HBM peak — a candidate where every lookup is a load
        #    (grid marchers, voxel-sized steps) or a BVH read;
        #  - with the committed counters of this workload, two more candidates: "hbm-latency" — the HBM traffic of random
        #    4-byte gathers (each drags a 128-byte line in) against the peak — and "valu" — the instruction rate against the
        #    issue peak;
        #  - without counters (another sample count, N > 1) and lookups served from registers: the loaded bytes against the HBM peak.
        #  With counters the named bound is the one the kernel sits closest to (the largest of the candidate fractions).
        cands = []
        if mesh_scene or loads_per_lookup >= 0.5:
            cands.append(("hbm", achieved, peak_gbs, "GB/s"))
        if hbm_obj and valu_obj:
            cands.append(("hbm-latency", hbm_obj["achieved"], peak_gbs, "GB/s"))
            cands.append(("valu", valu_obj["achieved"], valu_obj["peak"], valu_obj["unit"]))
        if not cands:
            cands.append(("hbm", loads_obj["achieved"], peak_gbs, "GB/s"))
        bound, r_ach, r_peak, r_unit = max(cands, key=lambda c: c[1] / c[2])
        roofline = {
            "bound": bound, "achieved": r_ach, "peak": r_peak, "unit": r_unit, "frac": r_ach / r_peak, "traffic": traffic,
            "kernel": kernel_name, "kernel_ms_per_launch": avg_launch_s * 1e3, "launches": int(launches_all),
            "algorithmic_bytes_per_launch": bytes_per_launch, "lookup_rate": lookup_rate,
            "note": "frac is the fraction of the bound named in `bound`; `lookup_rate` is SURVEY 8(d)'s algorithmic figure (a rate, not a "
                    "fraction, wherever lookups are served without memory), `loads` what the memory system is asked for, `hbm` what HBM "
                    "delivered (committed PMC passes), `valu` the instruction rate (DESIGN.md 6)",
        }
        if loads_obj:
            roofline["loads"] = loads_obj
        if hbm_obj:
            roofline["hbm"] = hbm_obj
        if valu_obj:
            roofline["valu"] = valu_obj
        out = {
            "metric": "Mpath-samples/sec at 1080p (1/2/4/8 GPU) + density lookups/s vs HBM roofline",
            "value": value,
            "unit": "Mpath-samples/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": a.scaling if world > 1 else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": a.workload, "integrator": integ, "image": f"{W}x{H}", "spp": int(opts.rayPerPixelCount), "spp_at_1_gpu": spp,
                "grid": (f"{n} mesh, threaded BVH" if mesh_scene else
                         f"{n}^3 procedural cloud (Perlin fbm), bricked 8^3, replicated per GPU" +
                         (", bounded by a 320-triangle sphere (threaded BVH)" if mesh_tracker else "")), "step_regime": regime,
                "kernel_mode": a.mode,
                "partition": ("REHEARSAL: all ranks on one GPU, gloo" if rehearsal else f"rows r::{world} per rank, RCCL all_gather") if world > 1 else "single GPU",
                "timed_span": "render + gather + frame copied to host memory on rank 0",
                "frame_mean_rgb": frame_mean,
            },
            "glookups_per_s": lookups_all / max(kernel_ms_all / 1e3 / world, 1e-12) / 1e9,
            "lookups_per_sample": lookups_all / (samples_per_step * a.steps),
            "divergence": {k: stats[k] for k in ("wave_trips", "wave_trip_lanes", "event_phases", "scatter_events")}
            | {"walking_lanes_per_trip": stats["wave_trip_lanes"] / max(stats["wave_trips"], 1)},
            "roofline": roofline,
        }
        if weak is not None:
            out["weak_scaling"] = weak
        if piped is not None:
            out["two_frames_in_flight"] = piped
        want_cpu = a.cpu_baseline == "on" or (a.cpu_baseline == "auto" and world == 1)
        if want_cpu:
            try:
                out["cpu_baseline"] = cpu_baseline(integ_id, n, opts, a.cpu_seconds, mesh,
                                                   gpu_frame=sched.frame.cpu().numpy() if world == 1 and weak is None else None,
                                                   local=a.mode == "local")
            except Exception as e:  # the baseline is reported, never required for the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "Mpath-samples/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
