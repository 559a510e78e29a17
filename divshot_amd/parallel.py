"""Data-parallel sharding of training views across GPUs.

The reference is single-GPU (no NCCL/MPI call sites anywhere in its tree); this is new functionality:
one process per GPU, every rank holds a full replica of the splat parameter block (236 B/splat), views are independent
units sharded round-robin, and the only exchange step is a sum-all-reduce of the 59-float gradient rows. The rows of
all six parameter groups live in ONE flat fp32 buffer so the exchange is a single large collective (236 MB at 1M splats):
on MI355X's point-to-point xGMI mesh a few large messages use the links far better than many small ones.
Backend: "nccl" (= RCCL) on GPUs, "gloo" in the CPU tests.
"""
import torch

PARAM_KEYS = ("pos", "sh0", "shN", "opacity", "scale", "rot")
PARAM_WIDTH = {"pos": 3, "sh0": 3, "shN": 45, "opacity": 1, "scale": 3, "rot": 4}
ROW_FLOATS = sum(PARAM_WIDTH.values())      # 59


def views_for_rank(n_views, rank, world):
    """Round-robin view shard: rank r renders views r, r+world, ... (weak scaling when n_views == world)."""
    return list(range(rank, n_views, world))


class GradBuffer:
    """Flat [n*59] gradient buffer with per-group views shaped like the parameter arrays."""

    def __init__(self, n, device):
        self.n = n
        self.flat = torch.zeros(n * ROW_FLOATS, dtype=torch.float32, device=device)
        self.views, off = {}, 0
        shapes = {"pos": (n, 3), "sh0": (n, 3), "shN": (n, 15, 3), "opacity": (n,), "scale": (n, 3), "rot": (n, 4)}
        for k in PARAM_KEYS:
            cnt = n * PARAM_WIDTH[k]
            self.views[k] = self.flat[off:off + cnt].view(shapes[k])
            off += cnt
        assert off == self.flat.numel()

    def all_reduce(self, group=None, average=False):
        """Sum (or mean) the gradient rows over all ranks. No-op without an initialised process group."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return None
        work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            self.flat /= dist.get_world_size(group)
        return work
