"""bench.py with more than one rank, rehearsed on the one GPU of the test box: every rank on GPU 0, the exchange
over gloo (VR_BENCH_REHEARSAL=1; RCCL refuses two ranks on one device).  What is checked is the N > 1 plumbing —
rendezvous, row sets, gather, max-over-ranks timing, the JSON line — not a rate."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(nranks, *args):
    env = dict(os.environ, VR_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    if nranks == 1:
        cmd = [sys.executable, "bench.py"]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nranks}",
               "--master-addr", "127.0.0.1", "--master-port", str(29600 + nranks), "bench.py"]
    cmd += ["--gpus", str(nranks), "--steps", "1", "--warmup", "0", "--workload", "small", "--cpu-baseline", "off", *args]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # ONE json line, from rank 0
    return json.loads(lines[0])


def test_two_ranks_weak_and_strong():
    one = run_bench(1, "--spp", "8")
    weak = run_bench(2, "--scaling", "weak")           # small is 4 spp: two ranks render 8
    strong = run_bench(2, "--spp", "8")                # the default: the named frame split by rows, weak measured beside it
    assert strong["weak_scaling"]["spp"] == 16 and strong["weak_scaling"]["value"] > 0 and "weak_scaling" not in one
    assert one["scaling"] == "strong" and one["config"]["timed_span"].endswith("host memory on rank 0")
    assert one["roofline"]["loads"]["bytes_per_launch"] <= one["roofline"]["algorithmic_bytes_per_launch"]
    assert weak["n_gpus"] == 2 and weak["scaling"] == "weak" and weak["config"]["spp"] == 8 and weak["config"]["spp_at_1_gpu"] == 4
    assert strong["scaling"] == "strong" and strong["config"]["spp"] == 8
    # the frame does not depend on how many ranks rendered it (streams keyed by absolute pixel and sample)
    assert weak["config"]["frame_mean_rgb"] == one["config"]["frame_mean_rgb"] == strong["config"]["frame_mean_rgb"]
    assert weak["lookups_per_sample"] == one["lookups_per_sample"]
    for d in (one, weak, strong):
        assert d["value"] > 0 and d["roofline"]["launches"] == d["n_gpus"]


def test_two_real_gpus_rccl_and_peer_copies():
    """On a box with at least two GPUs (the round-end node; the one-GPU test box skips): bench.py's N = 2 path over RCCL
    without the rehearsal switch, and one frame over two devices behind the C-ABI (peer copies into the first GPU)."""
    import numpy as np
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("VR_BENCH_REHEARSAL", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29655", "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "1", "--workload", "small", "--cpu-baseline", "off"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    two = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert two["n_gpus"] == 2 and "RCCL" in two["config"]["partition"] and two["weak_scaling"]["value"] > 0
    from volume_renderer_amd import MODE_FAST, Backend, BackendGroup
    from volume_renderer_amd import workloads as wl
    o = wl.options_cloud("ratioTracking", 64, 96, 50, 4)
    with Backend(0) as one:
        one.generate_cloud(64, wl.cloud_radius(64)); one.set_seed(wl.SEED); one.set_mode(MODE_FAST)
        want = one.render(o, "ratioTracking")
    with BackendGroup([0, 1]) as g:
        g.generate_cloud(64, wl.cloud_radius(64)); g.set_seed(wl.SEED); g.set_mode(MODE_FAST)
        assert np.array_equal(g.render(o, "ratioTracking"), want)
        frame = torch.zeros((50, 96, 3), dtype=torch.float32, device="cuda:0")
        torch.cuda.synchronize()
        g.render_device(o, "ratioTracking", frame.data_ptr())
        assert np.array_equal(frame.cpu().numpy(), want)
