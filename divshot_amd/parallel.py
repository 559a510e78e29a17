"""Data-parallel sharding of training views across GPUs.

The reference is single-GPU (no NCCL/MPI call sites anywhere in its tree); this is new functionality:
one process per GPU, every rank holds a full replica of the splat parameter block (236 B/splat), views are independent
units sharded round-robin, and the only exchange step is a sum-all-reduce of the 59-float gradient rows. The rows of
all six parameter groups live in ONE flat fp32 buffer so the exchange is a single large collective (236 MB at 1M splats):
on MI355X's point-to-point xGMI mesh a few large messages use the links far better than many small ones.
Backend: "nccl" (= RCCL) on GPUs, "gloo" in the CPU tests.
"""
import os
import torch

# DVS_FORCE_COLLECTIVES=1: issue every collective even in a 1-rank group (RCCL allows a 1-rank communicator) — lets a single-GPU
# box prove librccl loading, communicator creation and the side-stream collectives before the first multi-GPU run
def _force():
    return os.environ.get("DVS_FORCE_COLLECTIVES", "0") == "1"


PARAM_KEYS = ("pos", "sh0", "shN", "opacity", "scale", "rot")
PARAM_WIDTH = {"pos": 3, "sh0": 3, "shN": 45, "opacity": 1, "scale": 3, "rot": 4}
ROW_FLOATS = sum(PARAM_WIDTH.values())      # 59
# order inside the flat buffer: the 11 "geometry" floats first (always all-reduced), then the 48 SH floats
FLAT_ORDER = ("pos", "scale", "rot", "opacity", "sh0", "shN")
GEOM_FLOATS = 3 + 3 + 4 + 1                 # 11 floats = 44 B per splat
SHAPES = {"pos": (-1, 3), "sh0": (-1, 3), "shN": (-1, 15, 3), "opacity": (-1,), "scale": (-1, 3), "rot": (-1, 4)}


def views_for_rank(n_views, rank, world):
    """Round-robin view shard: rank r renders views r, r+world, ... (weak scaling when n_views == world)."""
    return list(range(rank, n_views, world))


class GradBuffer:
    """Flat [n*59] gradient buffer with per-group views shaped like the parameter arrays.
    flat_geom = the leading 11 floats per splat (pos, scale, rot, opacity); flat_sh = the trailing 48 (sh0, shN)."""

    def __init__(self, n, device, shn_tiled=False):
        self.n = n
        shn_floats = ((n + 63) // 64) * 64 * 48 if shn_tiled else n * 45     # DVS_SHN_TILED: whole 64-splat tiles, 48 floats/splat
        # every group starts on a 16-byte boundary (the kernels move rot / shN / the 3-float groups as 16-B vectors, dvs_raster.h)
        counts = {k: (shn_floats if k == "shN" else n * PARAM_WIDTH[k]) for k in FLAT_ORDER}
        offs, off = {}, 0
        for k in FLAT_ORDER:
            offs[k] = off
            off += (counts[k] + 3) & ~3
        self.flat = torch.zeros(off, dtype=torch.float32, device=device)
        # floats per group INCLUDING its alignment pad, in flat order: what an element-wise consumer of `flat` (ShardedAdam) must be
        # built from — sum(group_sizes) == flat.numel(); a group's learning rate then also covers its (always zero) pad floats
        self.group_sizes = [(((counts[k] + 3) & ~3)) for k in FLAT_ORDER]
        self.group_offsets = [offs[k] for k in FLAT_ORDER]
        self.views = {}
        for k in FLAT_ORDER:
            v = self.flat[offs[k]:offs[k] + counts[k]]
            self.views[k] = v if (k == "shN" and shn_tiled) else v.view([n if d == -1 else d for d in SHAPES[k]])
        self.flat_geom = self.flat[: offs["sh0"]]          # pos, scale, rot, opacity (+ pad floats, always zero)
        self.flat_sh = self.flat[offs["sh0"]:]

    def all_reduce(self, group=None, average=False):
        """Sum (or mean) ALL gradient rows over all ranks (236 B/splat on the wire). No-op without a process group."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not _force()):
            return None
        work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            self.flat /= dist.get_world_size(group)
        return work


class FactorisedExchange:
    """Gradient exchange that ships 56 B/splat instead of 236 B/splat.

    The SH gradient rows of one view are rank-1 in that view's 3-float colour gradient:
        dL/dsh0 = SH_C0 * gc,   dL/dshN[k] = basis_k(normalize(pos - campos_view)) * gc
    (preprocess backward). So instead of all-reducing 48 SH floats per splat, every rank
    all-gathers the other ranks' gc (3 floats per splat per view) and rebuilds the summed SH rows locally with
    dvs_sh_grad_combine, from its own replica of the positions and the views' camera centres. Only the 11 geometry
    floats (pos, scale, rot, opacity) go through the sum-all-reduce. Per-GPU wire volume at 8 GPUs and 1M splats:
    ring all-reduce of 236 MB ~ 413 MB  ->  all-reduce of 44 MB (77 MB) + per view an all-gather of 8 x 12 MB (84 MB received).
    With several views per rank and step, the gather of view v runs on a side stream under the compute of view v+1
    (gather_view), so the exposed part stays one gather + the geometry all-reduce however many views a step has.
    On MI355X's point-to-point xGMI links the exchange is bandwidth-bound, so this is ~2.5x less exposed time.
    """

    def __init__(self, n, device, world, views_per_rank=1, rank_major=False):
        """rank_major: dcolor_all[r * views_per_rank + v] = view v of rank r — the layout ONE all-gather of all local views produces
        (gather_all); default view-major, one all-gather per view (gather_view)."""
        self.n, self.world, self.views_per_rank, self.rank_major = n, world, views_per_rank, rank_major
        # dcolor_local[v] = colour gradient of this rank's v-th view; after the all-gathers dcolor_all[v*world + r] = view v of
        # rank r (view-major, so the slots of one view are one contiguous all-gather output; `slots()` lists the order)
        self.dcolor_local = torch.zeros((views_per_rank, n, 3), dtype=torch.float32, device=device)
        self.dcolor_all = torch.zeros((world * views_per_rank, n, 3), dtype=torch.float32, device=device)
        self._gathered = [False] * views_per_rank
        self._comm = torch.cuda.Stream(device=device) if torch.device(device).type == "cuda" else None
        self._combiner = None
        self._n_combined = 0
        self._geom_reduced = False        # the geometry groups went out chunk by chunk (reduce_geometry_chunk): communicate() skips the big one

    def reduce_geometry_chunk(self, gbuf, first, count, ready=None, group=None):
        """"launch the reduce for splat-chunk k as soon as A9 has finished chunk k". Sum-all-reduce of the four
        geometry groups' rows [first, first + count) — 44 B/splat, four contiguous ranges of the flat buffer — on the exchange's side stream behind `ready` (an event recorded after the chunk's A9
        launch): it runs under the A9 of the following chunks, so only the last chunk's reduce stays exposed."""
        import torch.distributed as dist
        self._geom_reduced = True
        if self.world == 1 and not (_force() and dist.is_initialized()):
            return
        parts = [gbuf.views[k][first:first + count] for k in ("pos", "scale", "rot", "opacity")]
        def run():          # (four collectives per chunk: torch has no public grouped launch; libgstrain.so issues them as ONE ncclGroup)
            for t in parts:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        if self._comm is None:
            run()
        else:
            if ready is not None:
                self._comm.wait_event(ready)
            else:
                self._comm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._comm):
                run()

    def set_combiner(self, fn):
        """fn(slot_lo, slot_hi, accumulate): rebuild the SH rows of slots [slot_lo, slot_hi) (dvs_sh_grad_combine) — when set, the
        rows of view v's slots are rebuilt right behind its all-gather on the side stream (nothing else writes the SH rows in
        factorised mode), so only the last view's gather + rebuild stay exposed at the end of the step."""
        self._combiner = fn

    def slots(self):
        """[(rank, local_view)] for every slot of dcolor_all, in order — index the per-view camera table with this."""
        if self.rank_major:
            return [(r, v) for r in range(self.world) for v in range(self.views_per_rank)]
        return [(r, v) for v in range(self.views_per_rank) for r in range(self.world)]

    def gather_all(self, ready=None, group=None):
        """rank_major layout: ONE all-gather of all local views' colour gradients, started on the side stream as soon as `ready`
        (a CUDA event recorded behind dvs_raster_backward_dcolor) has passed — it runs under the preprocess backward (A9), so only
        the geometry all-reduce stays exposed at the end of the step."""
        import torch.distributed as dist
        assert self.rank_major
        def run():
            if self.world == 1 and not (_force() and dist.is_initialized()):
                self.dcolor_all.copy_(self.dcolor_local)
                return
            try:
                dist.all_gather_into_tensor(self.dcolor_all.view(-1), self.dcolor_local.view(-1), group=group)
            except (RuntimeError, NotImplementedError):
                per = self.views_per_rank
                dist.all_gather([self.dcolor_all[r * per:(r + 1) * per] for r in range(self.world)], self.dcolor_local, group=group)
        if self._comm is None:
            run()
        else:
            if ready is not None:
                self._comm.wait_event(ready)
            else:
                self._comm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._comm):
                run()
        self._gathered = [True] * self.views_per_rank

    def _gather(self, v, group):
        import torch.distributed as dist
        out = self.dcolor_all[v * self.world:(v + 1) * self.world]
        if self.world == 1 and not (_force() and dist.is_initialized()):
            out[0].copy_(self.dcolor_local[v])
            return
        try:
            dist.all_gather_into_tensor(out.view(-1), self.dcolor_local[v].view(-1), group=group)
        except (RuntimeError, NotImplementedError):      # backends without the flat form (gloo on some builds)
            dist.all_gather(list(out.unbind(0)), self.dcolor_local[v], group=group)

    def gather_view(self, v, ready=None, group=None):
        """Start the all-gather of local view v as soon as its backward has produced dcolor_local[v] (`ready`: a CUDA event
        recorded after that backward). It runs on a side stream, under the composite kernels of the following views, so only
        the last view's gather and the small geometry all-reduce stay exposed at the end of the step."""
        if self._comm is None:
            self._gather(v, group)
            if self._combiner is not None:
                self._combiner(v * self.world, (v + 1) * self.world, self._n_combined > 0)
                self._n_combined += 1
        else:
            if ready is not None:
                self._comm.wait_event(ready)
            else:
                self._comm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._comm):
                self._gather(v, group)
                if self._combiner is not None:
                    self._combiner(v * self.world, (v + 1) * self.world, self._n_combined > 0)
                    self._n_combined += 1
        self._gathered[v] = True

    def communicate(self, gbuf, group=None):
        """all-gather of the views not gathered yet + all-reduce(geometry slice). Complete on the current stream on return."""
        import torch.distributed as dist
        if self.rank_major:
            if not all(self._gathered):
                self.gather_all(None, group)
        else:
            for v in range(self.views_per_rank):
                if not self._gathered[v]:
                    self.gather_view(v, None, group)
        self._gathered = [False] * self.views_per_rank
        self._n_combined = 0
        if not self._geom_reduced and (self.world > 1 or (_force() and dist.is_initialized())):
            dist.all_reduce(gbuf.flat_geom, op=dist.ReduceOp.SUM, group=group)
        self._geom_reduced = False
        if self._comm is not None:
            torch.cuda.current_stream().wait_stream(self._comm)

    def exchange(self, gbuf, rast, pos, campos_all, sh_degree, group=None, shn_tiled=False):
        """Full exchange: after this, every view of gbuf holds the sum over all ranks' views. campos_all[s] is the camera
        centre of slot s (see slots())."""
        self.communicate(gbuf, group)
        if self._combiner is None:          # otherwise the rows were rebuilt slot by slot behind the gathers
            rast.sh_grad_combine(pos, campos_all, self.dcolor_all, gbuf.views["sh0"], gbuf.views["shN"], sh_degree, shn_tiled=shn_tiled)


class ShardedAdam:
    """The optimizer half of the data-parallel step in its bandwidth-optimal form:
        reduce-scatter(gradient rows)  ->  Adam on this rank's 1/G slice of the flat buffer  ->  all-gather(parameters).
    Same wire volume as one all-reduce of the gradients, but the Adam work and its two moment arrays (472 B/splat) are divided
    by G instead of being replicated. Parameters and gradients live in flat buffers in GradBuffer's order, so a rank's slice is
    one contiguous range that may straddle parameter groups; Adam is element-wise, the per-group learning rates are applied to
    the sub-ranges (dvs_adam_step_groups, one launch). Backends without reduce_scatter (gloo) fall back to all-reduce + slice.
    """

    def __init__(self, params_flat, group_sizes, lrs, world, rank, beta1=0.9, beta2=0.999, eps=1e-15, group=None):
        """params_flat: flat fp32 tensor; group_sizes: floats per group in flat order; lrs: learning rate per group."""
        assert sum(group_sizes) == params_flat.numel() and len(lrs) == len(group_sizes)
        self.p, self.world, self.rank, self.group = params_flat, world, rank, group
        self.betas, self.eps, self.lrs = (beta1, beta2), eps, list(lrs)
        n = params_flat.numel()
        self.shard = (n + world - 1) // world
        self.shard = (self.shard + 3) & ~3                        # 16-B aligned slices
        self.padded = self.shard * world
        dev = params_flat.device
        self.lo = min(n, rank * self.shard)
        self.hi = min(n, self.lo + self.shard)
        self.m = torch.zeros(self.shard, dtype=torch.float32, device=dev)      # moments only for the own slice
        self.v = torch.zeros(self.shard, dtype=torch.float32, device=dev)
        self.gshard = torch.zeros(self.shard, dtype=torch.float32, device=dev)
        self.pshard = torch.zeros(self.shard, dtype=torch.float32, device=dev)
        self._gpad = torch.zeros(self.padded, dtype=torch.float32, device=dev) if self.padded != n else None
        self._ppad = torch.zeros(self.padded, dtype=torch.float32, device=dev) if self.padded != n else None
        # sub-ranges of the own slice per parameter group: (offset in slice, count, lr)
        self.ranges, off = [], 0
        for size, lr in zip(group_sizes, lrs):
            a, b = max(off, self.lo), min(off + size, self.hi)
            if b > a:
                self.ranges.append((a - self.lo, b - a, lr))
            off += size
        self.step_no = 0

    def _reduce_scatter(self, grads_flat):
        import torch.distributed as dist
        src = grads_flat
        if self._gpad is not None:
            self._gpad[: grads_flat.numel()].copy_(grads_flat)
            src = self._gpad
        if self.world == 1 and not (_force() and dist.is_initialized()):
            self.gshard.copy_(src[: self.shard])
            return
        try:
            dist.reduce_scatter_tensor(self.gshard, src, op=dist.ReduceOp.SUM, group=self.group)
        except (RuntimeError, NotImplementedError, ValueError):         # gloo: no reduce-scatter
            dist.all_reduce(src, op=dist.ReduceOp.SUM, group=self.group)
            self.gshard.copy_(src[self.rank * self.shard:(self.rank + 1) * self.shard])

    def _adam(self, lr_scale=1.0):
        b1, b2 = self.betas
        t = self.step_no
        def torch_adam(o, c, lr):                                      # the textbook recurrences on a sub-range, in torch
            g = self.gshard[o:o + c]
            self.m[o:o + c].mul_(b1).add_(g, alpha=1 - b1)
            self.v[o:o + c].mul_(b2).addcmul_(g, g, value=1 - b2)
            mh = self.m[o:o + c] / (1 - b1 ** t); vh = self.v[o:o + c] / (1 - b2 ** t)
            self.pshard[o:o + c].sub_(lr * lr_scale * mh / (vh.sqrt() + self.eps))
        if self.p.is_cuda:
            from .train_ops import adam_step_groups
            groups = []
            for o, c, lr in self.ranges:
                head = min(c, (-o) % 4)                                # dvs_adam_step_groups moves float4: the kernel's part of a sub-range
                if head:                                               # starts on a 16-byte boundary, the <= 3 floats before it go through torch
                    torch_adam(o, head, lr)
                if c > head:
                    o2, c2 = o + head, c - head
                    groups.append(dict(param=self.pshard[o2:o2 + c2], grad=self.gshard[o2:o2 + c2], m=self.m[o2:o2 + c2], v=self.v[o2:o2 + c2],
                                       lr=lr * lr_scale, width=1))
            for i in range(0, len(groups), 8):
                adam_step_groups(groups[i:i + 8], t, b1, b2, self.eps)
        else:                                                           # CPU tensors (gloo tests)
            for o, c, lr in self.ranges:
                torch_adam(o, c, lr)

    def step(self, grads_flat, lr_scale=1.0):
        """One optimizer step on every rank's replica: after it, params_flat holds the updated parameters everywhere."""
        import torch.distributed as dist
        if grads_flat.numel() != self.p.numel():
            raise ValueError(f"ShardedAdam.step: gradient buffer has {grads_flat.numel()} floats, the parameters {self.p.numel()} "
                             "(build both from GradBuffer.group_sizes)")
        self.step_no += 1
        n = self.p.numel()
        self._reduce_scatter(grads_flat)
        self.pshard[: self.hi - self.lo].copy_(self.p[self.lo:self.hi])
        self._adam(lr_scale)
        if self.world == 1 and not (_force() and dist.is_initialized()):
            self.p.copy_(self.pshard[:n])
            return
        dst = self._ppad if self._ppad is not None else self.p
        try:
            dist.all_gather_into_tensor(dst, self.pshard, group=self.group)
        except (RuntimeError, NotImplementedError, ValueError):
            parts = [torch.empty_like(self.pshard) for _ in range(self.world)]
            dist.all_gather(parts, self.pshard, group=self.group)
            torch.cat(parts, out=dst)
        if self._ppad is not None:
            self.p.copy_(self._ppad[:n])


# ---- the product's exchange: include/dvs_comm.h (librccl behind plain C), driven with the plugin's stream / event choreography ---------
class DvsComm:
    """ctypes handle of a dvs_comm communicator (include/dvs_comm.h): the SAME C entry points libgstrain.so's train_step() calls
    (divshot_amd/gstrain/gstrain.cpp) — RCCL over xGMI, or, with DVS_COMM_BACKEND=tcp, the host-staged test backend. rank / world /
    rendezvous come from the launcher's environment (RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT) unless given."""

    def __init__(self, device_index, rank=-1, world=0, master_addr=None, master_port=0):
        from ._lib import lib, DvsError
        self._lib = lib
        self.h = lib.dvs_comm_create(int(device_index), int(rank), int(world), master_addr.encode() if master_addr else None, int(master_port))
        if not self.h:
            raise DvsError("dvs_comm_create failed: " + lib.dvs_last_error().decode())
        self.rank, self.world = lib.dvs_comm_rank(self.h), lib.dvs_comm_world(self.h)
        self.backend = lib.dvs_comm_backend_name(self.h).decode()
        self.backend_ranks = lib.dvs_comm_backend_ranks(self.h)        # what RCCL itself reports (ncclCommCount)
        self.dev = torch.device("cuda", device_index)
        self._scratch_i = torch.zeros(4, dtype=torch.int32, device=self.dev)
        self._scratch_f = torch.zeros(max(4, self.world), dtype=torch.float32, device=self.dev)

    def close(self):
        if self.h:
            self._lib.dvs_comm_destroy(self.h)
            self.h = None

    @staticmethod
    def _sp(stream):
        import ctypes as C
        return C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)

    def _ok(self, status, what):
        if status != 0:
            from ._lib import DvsError
            raise DvsError(f"{what} failed with status {status}: {self._lib.dvs_last_error().decode()}")

    def all_reduce_sum(self, t, stream=None):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        self._ok(self._lib.dvs_comm_all_reduce_sum_f32(self.h, self._sp(stream), t.data_ptr(), t.numel()), "dvs_comm_all_reduce_sum_f32")

    def all_gather(self, send, recv, stream=None):
        assert send.is_cuda and recv.is_cuda and send.dtype == recv.dtype == torch.float32 and recv.numel() == send.numel() * self.world
        self._ok(self._lib.dvs_comm_all_gather_f32(self.h, self._sp(stream), send.data_ptr(), recv.data_ptr(), send.numel()), "dvs_comm_all_gather_f32")

    def group_start(self):
        self._ok(self._lib.dvs_comm_group_start(self.h), "dvs_comm_group_start")

    def group_end(self):
        self._ok(self._lib.dvs_comm_group_end(self.h), "dvs_comm_group_end")

    # host-visible helpers of a benchmark (each synchronises the current stream)
    def barrier(self):
        self._ok(self._lib.dvs_comm_all_reduce_max_i32(self.h, self._sp(None), self._scratch_i.data_ptr(), 1), "dvs_comm_all_reduce_max_i32")
        torch.cuda.current_stream().synchronize()

    def max_over_ranks(self, seconds):
        """max of a host float over the ranks (as integer microseconds through the int32 max-all-reduce)."""
        self._scratch_i[0] = int(min(2 ** 31 - 1, round(seconds * 1e6)))
        self._ok(self._lib.dvs_comm_all_reduce_max_i32(self.h, self._sp(None), self._scratch_i.data_ptr(), 1), "dvs_comm_all_reduce_max_i32")
        torch.cuda.current_stream().synchronize()
        return int(self._scratch_i[0].item()) * 1e-6

    def gather_floats(self, value):
        """[value of rank 0, ..., value of rank world - 1] on every rank."""
        send = torch.full((1,), float(value), dtype=torch.float32, device=self.dev)
        recv = self._scratch_f[: self.world]
        self.all_gather(send, recv)
        torch.cuda.current_stream().synchronize()
        return [float(x) for x in recv.cpu()]


class DvsCommExchange:
    """The factorised exchange of FactorisedExchange, executed by the PRODUCT's communication layer with the product's choreography
    (gstrain.cpp train_step): every collective goes through include/dvs_comm.h on a dedicated communication stream,
    in the same order on every rank —
      * the colour gradients of all local views leave in ONE all-gather as soon as dvs_raster_backward_dcolor has produced them
        (under A9),
      * A9 may run in splat chunks; chunk k's 44 B/splat of geometry gradients (four ranges of the flat buffer) leave as ONE grouped
        launch behind its event, under the A9 of the chunks behind it; otherwise one all-reduce of the geometry prefix behind A9,
      * the SH rows are rebuilt on the compute stream as soon as the all-gather has landed — while the geometry all-reduce is still
        on the links — and the step ends when that has landed too.
    Same attribute / method names as FactorisedExchange's rank-major form, so bench.py's step drives either (--exchange-impl)."""
    rank_major = True

    def __init__(self, n, device, comm, views_per_rank=1):
        self.n, self.comm, self.world, self.views_per_rank = n, comm, comm.world, views_per_rank
        self.dcolor_local = torch.zeros((views_per_rank, n, 3), dtype=torch.float32, device=device)
        self.dcolor_all = torch.zeros((self.world * views_per_rank, n, 3), dtype=torch.float32, device=device)
        self._comm = torch.cuda.Stream(device=device)
        self._ev_gather, self._ev_bwd, self._ev_comm = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()
        self._gathered = False
        self._geom_reduced = False

    def slots(self):
        return [(r, v) for r in range(self.world) for v in range(self.views_per_rank)]

    def gather_all(self, ready=None, group=None):
        if ready is not None:
            self._comm.wait_event(ready)
        else:
            self._comm.wait_stream(torch.cuda.current_stream())
        self.comm.all_gather(self.dcolor_local.view(-1), self.dcolor_all.view(-1), self._comm)
        self._ev_gather.record(self._comm)
        self._gathered = True

    def reduce_geometry_chunk(self, gbuf, first, count, ready=None, group=None):
        self._geom_reduced = True
        if ready is not None:
            self._comm.wait_event(ready)
        else:
            self._comm.wait_stream(torch.cuda.current_stream())
        self.comm.group_start()
        for k in ("pos", "scale", "rot", "opacity"):
            self.comm.all_reduce_sum(gbuf.views[k][first:first + count], self._comm)
        self.comm.group_end()

    def exchange(self, gbuf, rast, pos, campos_all, sh_degree, group=None, shn_tiled=False):
        cur = torch.cuda.current_stream()
        if not self._gathered:
            self.gather_all(None)
        if not self._geom_reduced:
            self._ev_bwd.record(cur)
            self._comm.wait_event(self._ev_bwd)
            self.comm.all_reduce_sum(gbuf.flat_geom, self._comm)
        self._ev_comm.record(self._comm)
        cur.wait_event(self._ev_gather)                 # the SH rows need only the colour all-gather: rebuilt under the geometry all-reduce
        rast.sh_grad_combine(pos, campos_all, self.dcolor_all, gbuf.views["sh0"], gbuf.views["shN"], sh_degree, shn_tiled=shn_tiled)
        cur.wait_event(self._ev_comm)
        self._gathered = False
        self._geom_reduced = False
