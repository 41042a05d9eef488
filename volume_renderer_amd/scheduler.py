"""Row scheduler across GPUs — the MI355X counterpart of the reference's chunk queue.

The reference hands 20-row chunks to CPU threads from a semaphore-guarded counter
(BaseIntegrator.cpp:253-285, Utils/MultithreadingHelper.cpp:10-28).  Across GPUs the split is
static and interleaved: rank r of N renders image rows r, r+N, r+2N, ...  Neighbouring rows
cost nearly the same, so every rank gets the same share of cheap (background) and expensive
(cloud core) rows without any run-time queue; inside a GPU the kernel's own work counter does
the dynamic balancing.  Every (pixel, sample) draws from a Philox stream keyed by its absolute
image coordinates, so the assembled frame is bit-identical for any N.

One exchange per frame: an all_gather of the per-rank row blocks (RCCL over xGMI with the
"nccl" backend; "gloo" in the CPU tests).  1080p is 24.9 MB in total, far below the size at
which link topology matters.
"""
import torch
import torch.distributed as dist


def rows_of_rank(rank, world, height):
    """(row_first, row_stride, row_count) of `rank`: rows rank::world of a height-row frame."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    return rank, world, len(range(rank, height, world))


def rows_per_rank_max(world, height):
    return (height + world - 1) // world


def assemble(gathered, height, world, out=None):
    """gathered: (world, rows_max, width, 3), rank-major -> (height, width, 3) frame."""
    _, _, width, ch = gathered.shape
    if out is None:
        out = torch.empty((height, width, ch), dtype=gathered.dtype, device=gathered.device)
    for r in range(world):
        k = len(range(r, height, world))
        out[r::world] = gathered[r, :k]
    return out


class RowScheduler:
    """Renders a frame across the ranks of the default process group.

    render_rows_fn(row_first, row_stride, row_count, out) must fill out[:row_count] (a
    (rows_max, width, 3) float32 tensor on this rank's device) with those rows and return when
    the data is complete (the HIP backend: render_row_set_device + synchronize).
    """

    def __init__(self, height, width, device, rank=None, world=None):
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.rank = rank if rank is not None else (dist.get_rank() if dist.is_initialized() else 0)
        self.height, self.width = height, width
        self.rows_max = rows_per_rank_max(self.world, height)
        self.part = torch.zeros((self.rows_max, width, 3), dtype=torch.float32, device=device)
        # all_gather_into_tensor concatenates along dim 0 (gloo insists on that shape)
        self.gathered = (torch.zeros((self.world * self.rows_max, width, 3), dtype=torch.float32, device=device)
                         if self.world > 1 else None)
        self.frame = torch.zeros((height, width, 3), dtype=torch.float32, device=device) if self.rank == 0 else None

    def render(self, render_rows_fn):
        first, stride, count = rows_of_rank(self.rank, self.world, self.height)
        render_rows_fn(first, stride, count, self.part)
        return self.gather()

    def gather(self):
        """The exchange alone: self.part (complete on this rank) -> the frame on rank 0.  Callers that keep several frames
        in flight render into `part` asynchronously themselves and call this once the rows are there."""
        if self.world > 1:
            dist.all_gather_into_tensor(self.gathered, self.part)
            if self.rank == 0:
                assemble(self.gathered.view(self.world, self.rows_max, self.width, 3), self.height, self.world,
                         out=self.frame)
        else:
            self.frame.copy_(self.part[:self.height])
        return self.frame

    def global_max(self, local_max):
        """Tonemap white point (BaseIntegrator.cpp:199-209) over all ranks' rows."""
        if self.world == 1:
            return float(local_max)
        t = torch.tensor([float(local_max)], dtype=torch.float32, device=self.part.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
