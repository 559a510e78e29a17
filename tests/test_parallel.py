"""N>1 path on CPU: two gloo processes exercise the view sharding and the flat-buffer gradient all-reduce that bench.py
runs over RCCL. The rasterizer itself needs a GPU; here each rank fills its rows with known values."""
import os
import socket
import sys
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n, out):
    sys.path.insert(0, ROOT)
    from divshot_amd.parallel import GradBuffer, views_for_rank, PARAM_KEYS
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gb = GradBuffer(n, torch.device("cpu"))
    total = {k: torch.zeros_like(gb.views[k]) for k in PARAM_KEYS}
    # every rank "renders" its shard of 8 views; view v contributes (v+1) * ramp to every group
    for v in views_for_rank(8, rank, world):
        for i, k in enumerate(PARAM_KEYS):
            gb.views[k] += (v + 1) * (i + 1) * torch.arange(gb.views[k].numel(), dtype=torch.float32).view(gb.views[k].shape) * 1e-3
    gb.all_reduce()
    for i, k in enumerate(PARAM_KEYS):
        want = sum(range(1, 9)) * (i + 1) * torch.arange(gb.views[k].numel(), dtype=torch.float32).view(gb.views[k].shape) * 1e-3
        assert torch.allclose(gb.views[k], want, rtol=1e-6), k
    out.put((rank, float(gb.flat.sum())))
    dist.destroy_process_group()


def _worker_fact(rank, world, port, n, out):
    sys.path.insert(0, ROOT)
    from divshot_amd.parallel import GradBuffer, FactorisedExchange
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gb = GradBuffer(n, torch.device("cpu"))
    V = 2                                                    # views per rank and step
    fx = FactorisedExchange(n, torch.device("cpu"), world, views_per_rank=V)
    ramp = torch.arange(n, dtype=torch.float32)[:, None]
    val = lambda r, v: (10 * v + r + 1) * torch.ones((n, 3)) + ramp
    for step in range(2):                                    # two steps: the per-step gather bookkeeping resets
        gb.flat.zero_()
        gb.flat_geom += (rank + 1) * torch.arange(gb.flat_geom.numel(), dtype=torch.float32) * 1e-3
        gb.flat_sh.fill_(-1.0)                               # must not be touched by the communication
        fx.dcolor_all.fill_(float("nan"))
        fx.dcolor_local[0].copy_(val(rank, 0))
        fx.gather_view(0)                                    # early gather of view 0 (on a GPU: under view 1's kernels)
        fx.dcolor_local[1].copy_(val(rank, 1))
        fx.communicate(gb)                                   # gathers view 1, all-reduces the geometry slice
        want = sum(range(1, world + 1)) * torch.arange(gb.flat_geom.numel(), dtype=torch.float32) * 1e-3
        assert torch.allclose(gb.flat_geom, want, rtol=1e-6)
        assert bool((gb.flat_sh == -1.0).all())
        assert fx.slots() == [(r, v) for v in range(V) for r in range(world)]
        for s_, (r, v) in enumerate(fx.slots()):             # view-major slots
            assert torch.equal(fx.dcolor_all[s_], val(r, v)), (s_, r, v)
    # rank-major layout: ONE all-gather of all local views (gather_all — what bench.py starts behind dvs_raster_backward_dcolor)
    fr = FactorisedExchange(n, torch.device("cpu"), world, views_per_rank=V, rank_major=True)
    for step in range(2):
        gb.flat.zero_()
        gb.flat_geom += (rank + 1) * torch.arange(gb.flat_geom.numel(), dtype=torch.float32) * 1e-3
        fr.dcolor_all.fill_(float("nan"))
        for v in range(V):
            fr.dcolor_local[v].copy_(val(rank, v))
        if step == 0:
            fr.gather_all()                                  # early; communicate() must not gather again
            fr.dcolor_local.fill_(float("nan"))
        fr.communicate(gb)                                   # step 1: communicate() does the gather itself
        want = sum(range(1, world + 1)) * torch.arange(gb.flat_geom.numel(), dtype=torch.float32) * 1e-3
        assert torch.allclose(gb.flat_geom, want, rtol=1e-6)
        assert fr.slots() == [(r, v) for r in range(world) for v in range(V)]
        for s_, (r, v) in enumerate(fr.slots()):
            assert torch.equal(fr.dcolor_all[s_], val(r, v)), (s_, r, v)
    out.put((rank, float(gb.flat_geom.sum())))
    dist.destroy_process_group()


def test_factorised_exchange_communication_gloo():
    world, n = 2, 193
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_fact, args=(r, world, port, n, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    sums = dict(out.get(timeout=5) for _ in range(world))
    assert abs(sums[0] - sums[1]) < 1e-3 * abs(sums[0])


def _worker_chunks(rank, world, port, n, out):
    sys.path.insert(0, ROOT)
    from divshot_amd.parallel import GradBuffer, FactorisedExchange
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gb = GradBuffer(n, torch.device("cpu"))
    fx = FactorisedExchange(n, torch.device("cpu"), world, views_per_rank=1, rank_major=True)
    fill = lambda r: (r + 1) * torch.arange(gb.flat_geom.numel(), dtype=torch.float32) * 1e-3
    want = sum(range(1, world + 1)) * torch.arange(gb.flat_geom.numel(), dtype=torch.float32) * 1e-3
    for step, chunks in enumerate(([(0, 256), (256, 256), (512, n - 512)], None, [(0, n)])):
        gb.flat.zero_(); gb.flat_geom += fill(rank); gb.flat_sh.fill_(-1.0)
        fx.dcolor_local[0].fill_(float(rank + step))
        if chunks is not None:                               # the geometry rows go out chunk by chunk (A9 of chunk k+1 runs above chunk k's reduce)
            for first, count in chunks:
                fx.reduce_geometry_chunk(gb, first, count)
        fx.communicate(gb)                                   # gathers dcolor; all-reduces the geometry slice only when no chunk went out
        assert torch.allclose(gb.flat_geom, want, rtol=1e-6), (step, "geometry summed once, not twice")
        assert bool((gb.flat_sh == -1.0).all())
        for r in range(world):
            assert bool((fx.dcolor_all[r] == float(r + step)).all())
    out.put((rank, float(gb.flat_geom.sum())))
    dist.destroy_process_group()


def test_chunked_geometry_reduce_equals_one_reduce_gloo():
    """SURVEY.md §8(e)'s chunked exchange on CPU: the four geometry groups reduced as three splat chunks (256-row boundaries, ragged
    tail) sum to what one all-reduce of the slice gives, communicate() does not reduce them a second time, and the flag resets."""
    world, n = 2, 700
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_chunks, args=(r, world, port, n, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    sums = dict(out.get(timeout=5) for _ in range(world))
    assert abs(sums[0] - sums[1]) < 1e-3 * abs(sums[0])


def test_view_sharding():
    from divshot_amd.parallel import views_for_rank
    for world in (1, 2, 4, 8):
        seen = sorted(v for r in range(world) for v in views_for_rank(8, r, world))
        assert seen == list(range(8))
        assert all(len(views_for_rank(8, r, world)) == 8 // world for r in range(world))


def test_grad_buffer_layout():
    from divshot_amd.parallel import GradBuffer, ROW_FLOATS
    gb = GradBuffer(12, torch.device("cpu"))                     # a multiple of 4 splats: no pad floats
    assert ROW_FLOATS == 59 and gb.flat.numel() == 12 * 59      # 59 fp32 = 236 B per splat (editor.cpp:1578)
    gb.views["rot"][3, 2] = 7.0
    assert gb.flat[12 * (3 + 3) + 3 * 4 + 2] == 7.0             # flat order: pos, scale, rot, opacity | sh0, shN
    assert gb.flat_geom.numel() == 12 * 11 and gb.flat_sh.numel() == 12 * 48
    gb.views["shN"][11, 14, 2] = 3.0
    assert gb.flat[-1] == 3.0
    assert gb.all_reduce() is None                              # no process group: single-GPU path is a no-op
    # any splat count: every group starts on a 16-byte boundary (the kernels move rot / shN / staged 3-float groups as 16-B vectors)
    for n in (1, 10, 257):
        g = GradBuffer(n, torch.device("cpu"))
        base = g.flat.data_ptr()
        for k, v in g.views.items():
            assert (v.data_ptr() - base) % 16 == 0, (n, k)
            assert v.numel() == n * {"pos": 3, "sh0": 3, "shN": 45, "opacity": 1, "scale": 3, "rot": 4}[k]
        assert g.flat_geom.numel() + g.flat_sh.numel() == g.flat.numel() and g.flat.numel() < n * 59 + 24


def test_two_rank_gradient_allreduce_gloo():
    world, n = 2, 257
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    sums = dict(out.get(timeout=5) for _ in range(world))
    assert abs(sums[0] - sums[1]) < 1e-3 * abs(sums[0])         # replicas hold identical reduced gradients


def _worker_sharded_adam(rank, world, port, n, out):
    sys.path.insert(0, ROOT)
    from divshot_amd.parallel import ShardedAdam
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sizes = [3 * n, 3 * n, 4 * n, n, 3 * n, 45 * n]                # flat order pos, scale, rot, opacity, sh0, shN
    lrs = [1e-2, 5e-3, 1e-3, 5e-2, 2.5e-3, 1.25e-4]
    total = sum(sizes)
    g0 = torch.Generator().manual_seed(0)
    p0 = torch.randn(total, generator=g0)
    params = p0.clone()
    opt = ShardedAdam(params, sizes, lrs, world, rank, eps=1e-8)
    # reference: replicated Adam on the summed gradient
    ref = p0.clone().double(); m = torch.zeros(total, dtype=torch.float64); v = torch.zeros(total, dtype=torch.float64)
    lr_vec = torch.cat([torch.full((s_,), lr, dtype=torch.float64) for s_, lr in zip(sizes, lrs)])
    for t in (1, 2, 3):
        gs = [torch.randn(total, generator=torch.Generator().manual_seed(100 * t + r)) for r in range(world)]
        opt.step(gs[rank].clone())
        gsum = sum(g.double() for g in gs)
        m = 0.9 * m + 0.1 * gsum; v = 0.999 * v + 0.001 * gsum * gsum
        ref = ref - lr_vec * (m / (1 - 0.9 ** t)) / ((v / (1 - 0.999 ** t)).sqrt() + 1e-8)
    assert torch.allclose(params.double(), ref, rtol=2e-5, atol=2e-6), float((params.double() - ref).abs().max())
    assert opt.m.numel() <= (total + world - 1) // world + 4          # moments are sharded, not replicated
    out.put((rank, float(params.sum())))
    dist.destroy_process_group()


def test_sharded_adam_gloo():
    """reduce-scatter -> Adam on the own slice -> all-gather == replicated Adam on the summed gradient."""
    world, n = 2, 37                                                # 59 * 37 = 2183 floats: slices straddle groups, padding exercised
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sharded_adam, args=(r, world, port, n, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    sums = dict(out.get(timeout=5) for _ in range(world))
    assert abs(sums[0] - sums[1]) < 1e-3                            # identical replicas after the all-gather


def test_bench_watchdog_turns_a_hang_into_an_error_line():
    """`python bench.py --gpus 2` whose ranks never come back (DVS_BENCH_TEST_HANG: they sleep before touching a GPU): after
    DVS_BENCH_WATCHDOG_S the run ends with ONE JSON line carrying "error" and value null, and a non-zero exit code — not with the
    caller's own 1800-s limit. Runs on CPU: the hang sits before any device work."""
    import json, subprocess, sys, time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(DVS_BENCH_TEST_HANG="1", DVS_BENCH_WATCHDOG_S="30")
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True,
                       text=True, timeout=240, env=env, cwd=ROOT)
    assert p.returncode != 0 and time.time() - t0 < 150
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-1000:] + p.stderr[-2000:]
    rec = json.loads(lines[0])
    assert rec["value"] is None and "watchdog" in rec["error"] and rec["n_gpus"] == 2
