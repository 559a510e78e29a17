#!/usr/bin/env python3
"""bench.py — train views/s (fwd+bwd raster) at 1M splats, 1920x1080, SH degree 3, on 1/2/4/8 MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 either launched by
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (one rank per GPU, RCCL) or started plainly, in which
case it launches its N ranks itself the same way.
A step = one training iteration's pass of the hot path over one batch of views = BASELINE.json config C4: 8 synthetic cameras per
iteration over the replicated 1M-splat scene, SHARDED over the GPUs (8/N views per GPU: rank r renders views r, r+N, ...), i.e.
strong scaling: the work of a step is fixed as N grows. Per view: A2..A7, dL/drgb = (rgb - target)/P, A8, A9, gradient rows
accumulated over the rank's views; the views of a rank are software-pipelined over two rasterizer contexts / HIP streams (steps do
not overlap). For N>1 the step ends by exchanging the 59-float gradient rows over xGMI: by default the
factorised exchange of divshot_amd/parallel.py (all-reduce of the 11 geometry floats + all-gather of the 3-float colour gradients,
SH rows rebuilt locally; 56 B/splat on the wire instead of 236), or with --exchange allreduce one sum-all-reduce of all rows.
value = 8 * K / time (whole job). `--views-per-step V` instead fixes V views per GPU per step (weak scaling, the round-1 shape).
Rank 0 prints ONE JSON line.  Inputs are resident in HBM before the timed region starts.
"""
import argparse
import datetime
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# multi-process GPU work on this driver stack needs dmabuf IPC (RCCL otherwise fails with hipIpcGetMemHandle: invalid argument); the
# variable is read when the HSA runtime starts, i.e. before anything below touches HIP
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

WORKLOADS = {
    # name: (n, W, H, sh_degree, scale_log_offset)
    "C3": (1_000_000, 1920, 1080, 3, 0.0),     # BASELINE.json configs[2] / metric config
    "C2": (100_000, 800, 800, 3, 0.0),
    "C1": (10_000, 256, 256, 0, 0.0),
    "C5": (5_000_000, 3840, 2160, 3, -math.log(2.0)),
}
HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def algorithmic_bytes(N, V, T, P, tiles, sh_degree, absgrad):
    """SURVEY.md §8(d) 'Algorithmic bytes per view', per stage."""
    K = (sh_degree + 1) ** 2
    B_sh = 12 * K
    p = math.ceil((32 + math.ceil(math.log2(max(tiles, 2)))) / 8)
    return {
        "preprocess_fwd": 44 * N + (B_sh + 48) * V + 8 * (N - V),
        "tile_scan": 8 * N,
        "duplicate": 20 * V + 12 * T,
        "sort": (8 + 24 * p) * T,
        "tile_ranges": 8 * T + 8 * tiles,
        "render_fwd": 40 * T + 20 * P,
        "render_bwd": 76 * T + 20 * P + (8 * T if absgrad else 0),
        "preprocess_bwd": (44 + B_sh + 36 + 48) * V + (44 + B_sh) * N,
    }, p


def read_clocks(dev_index=0):
    """Current shader / memory clock of the device (MHz) as the driver reports them: sysfs pp_dpm_* (the line marked '*'), else
    rocm-smi. SURVEY.md §8(d): 'clocks as found, report sclk/mclk'. Returns None entries when neither source is readable."""
    import glob, re, subprocess
    out = {"sclk_mhz": None, "mclk_mhz": None, "source": None}
    try:
        cards = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
        if cards:
            card = os.path.dirname(cards[min(dev_index, len(cards) - 1)])
            for key, fn in (("sclk_mhz", "pp_dpm_sclk"), ("mclk_mhz", "pp_dpm_mclk")):
                for line in open(os.path.join(card, fn)):
                    if "*" in line:
                        m = re.search(r"(\d+)\s*Mhz", line, re.I)
                        if m:
                            out[key] = int(m.group(1))
            out["source"] = "sysfs pp_dpm_sclk / pp_dpm_mclk (current level, read right after the timed region)"
    except Exception:      # noqa: BLE001
        pass
    if out["sclk_mhz"] is None:
        try:
            txt = subprocess.run(["rocm-smi", "-d", str(dev_index), "--showclocks"], capture_output=True, text=True, timeout=20).stdout
            for key, tag in (("sclk_mhz", "sclk"), ("mclk_mhz", "mclk")):
                m = re.search(tag + r" clock level.*?\((\d+)Mhz\)", txt, re.I)
                if m:
                    out[key] = int(m.group(1))
            out["source"] = "rocm-smi --showclocks"
        except Exception:      # noqa: BLE001
            pass
    return out


def kernel_source_sha():
    """sha256 (16 hex digits) over divshot_amd/csrc/*.{hip,h} (the device code and its headers) — the same digest tools/make_traffic_json.py stores in profiles/r*_traffic.json, so
    that the line can say whether the committed PMC counters were collected on the kernels it is timing (stale counters silently
    divided by fresh times)."""
    import glob, hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "divshot_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h"))):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def counters_provenance(tfile, tj):
    cur = kernel_source_sha()
    return {"file": "profiles/" + os.path.basename(tfile), "kernel_source_sha16": tj.get("kernel_source_sha16"), "current_kernel_source_sha16": cur,
            "collected_on_these_kernels": tj.get("kernel_source_sha16") == cur,
            "note": "counter bytes come from separate rocprofv3 --pmc passes committed under profiles/; when collected_on_these_kernels is false they "
                    "describe an earlier build and every *_by_counters figure mixes them with this run's times"}


def cpu_baseline(workload, max_seconds=60.0):
    """The CPU oracle (kind 'port': the reference's CPU libtorch path is not in its tree)
    timed on this box's host cores: one fwd+bwd view of the bench workload, all cores (OpenMP)."""
    import numpy as np
    import divshot_amd as dv
    from oracle import Oracle
    n, W, H, deg, soff = WORKLOADS[workload]
    cores = os.cpu_count() or 1
    # probe on a 1/16 sample of the splats at full resolution to bound the wall time
    frac = 1.0
    spec = dv.make_spec(n // 16, W, H, sh_degree=deg, scale_log_offset=soff)
    o = Oracle(np.float32)
    P = dv.synth_splats(spec); cam = dv.synth_camera(spec, 0); tgt = dv.synth_target(spec, 0)
    t0 = time.perf_counter(); img = o.forward(P, cam, sh_degree=deg); o.backward((img - tgt) / tgt[0].size)
    probe = time.perf_counter() - t0
    if probe * 16 > max_seconds:
        frac = 1.0 / 16
        secs, sample = probe, f"{workload} with 1/16 of the splats ({n // 16}) at {W}x{H}, 1 view fwd+bwd, OpenMP {cores} threads (full size predicted > {max_seconds:.0f}s)"
    else:
        spec = dv.make_spec(n, W, H, sh_degree=deg, scale_log_offset=soff)
        P = dv.synth_splats(spec); cam = dv.synth_camera(spec, 0); tgt = dv.synth_target(spec, 0)
        t0 = time.perf_counter(); img = o.forward(P, cam, sh_degree=deg); ref_grads = o.backward((img - tgt) / tgt[0].size)
        secs = time.perf_counter() - t0
        sample = f"{workload} full ({n} splats, {W}x{H}, SH{deg}), 1 view fwd+bwd, OpenMP {cores} threads"
        cpu_baseline.reference = {"img": img, "grads": ref_grads, "fragile": o.get("fragile").astype(bool),
                                  "num_rendered": int(o.get("vals").size)}          # the checker's outputs for view 0 (parity_vs_oracle)
    out = {"value": 1.0 / secs, "unit": "views/s (of the sample)", "cores": cores, "kind": "port",
           "sample": sample, "seconds": secs, "sample_fraction_of_splats": frac}
    # the same port on ONE core, on the smaller C2 config (100k splats, 800x800) so that it stays a few seconds (1 thread and all cores)
    try:
        from oracle import set_threads
        n2, W2, H2, deg2, soff2 = WORKLOADS["C2"]
        spec2 = dv.make_spec(n2, W2, H2, sh_degree=deg2, scale_log_offset=soff2)
        P2 = dv.synth_splats(spec2); cam2 = dv.synth_camera(spec2, 0); tgt2 = dv.synth_target(spec2, 0)
        res = {}
        for label, nthr in (("1_thread", 1), ("all_cores", 0)):
            set_threads(nthr)
            t0 = time.perf_counter(); img2 = o.forward(P2, cam2, sh_degree=deg2); o.backward((img2 - tgt2) / tgt2[0].size)
            res[label] = 1.0 / (time.perf_counter() - t0)
        set_threads(0)
        out["c2_views_per_s"] = {"workload": f"C2 ({n2} splats, {W2}x{H2}, SH{deg2}), 1 view fwd+bwd", **res}
    except Exception as e:      # noqa: BLE001
        out["c2_views_per_s"] = {"error": repr(e)}
    return out


cpu_baseline.reference = None


def scaling_model(n, world, views_per_rank, per_rank_compute_ms, measured_ms_per_step):
    """DESIGN.md §7's model of the data-parallel step, evaluated beside the measurement so that a scaling run tests a prediction:
    step = per-rank compute + exposed exchange. Factorised exchange over point-to-point xGMI with every peer link driven at once
    (direct reduce-scatter + all-gather): the colour all-gather moves views_per_rank * 12 B/splat per link and direction and overlaps A9
    (taken as 0.1 ms per local view); the all-reduce of the 44 B/splat geometry prefix moves 2 * 44/world B/splat per link and direction
    and is exposed. Assumed: 64 GB/s per link and direction achieved by RCCL, 30 us per collective. Inputs measured by this run: the
    ranks' compute time (start of a step to just before its exposed exchange). A MODEL, stated as such; null for one GPU."""
    if world <= 1 or not per_rank_compute_ms:
        return None
    bw, lat_ms = 64e9, 0.030
    t_gather = views_per_rank * n * 12 / bw * 1e3 + lat_ms
    t_reduce = 2 * (44 * n / world) / bw * 1e3 + lat_ms
    a9_ms = 0.1 * views_per_rank
    exposed = t_reduce + max(0.0, t_gather - a9_ms)
    compute = max(per_rank_compute_ms)
    step = compute + exposed
    # Second column (round 6): the PRODUCT's iteration (libgstrain's train_step = this raster step + the fused Adam) with the exchange
    # pipelined across the iteration boundary (DVS_EXCHANGE_PIPELINE=1, k = 4 A9 chunks); bench.py's step has no optimizer to hide anything
    # under, so this is not what the line measures. A small timeline, t = 0 at the start of A9 (the colour all-gather starts there):
    #   compute stream: A9 chunk j done at j * a9 / k; SH rebuild + SH Adam (0.25 ms per 10^6 splats: 192 of the 236 B) once the gather has
    #                   landed; then per chunk j, once ITS all-reduce has landed: geometry Adam + the next iteration's A2 of the chunk
    #                   ((0.06 + 0.086 * views) ms per 10^6 splats, / k)
    #   communication stream (serial): the gather, then all-reduce j (t_reduce / k + latency) as soon as A9 chunk j is done
    # exposed = end of the last chunk's work - what the same work takes without any communication.
    k = 4
    scale = n / 1e6
    w_sh, w_geo = 0.25 * scale, (0.06 + 0.086 * views_per_rank) * scale

    def product_exposed(chunks):
        t_chunk = t_reduce / chunks + lat_ms
        comm_free, ar_done = t_gather, []
        for j in range(1, chunks + 1):
            start = max(a9_ms * j / chunks, comm_free)
            comm_free = start + t_chunk
            ar_done.append(comm_free)
        t = max(a9_ms, t_gather) + w_sh
        if chunks == 1 or True:
            for j in range(chunks):
                t = max(t, ar_done[j]) + w_geo / chunks
        return t - (a9_ms + w_sh + w_geo)
    exposed_u, exposed_p = product_exposed(1), product_exposed(k)
    optimizer_ms = (0.22 + 0.06) * scale
    step_u, step_p = compute + optimizer_ms + exposed_u, compute + optimizer_ms + exposed_p
    return {"assumptions": {"link_GBps_per_direction": 64, "latency_us_per_collective": 30, "a9_ms_per_local_view": 0.1,
                            "links_driven": world - 1, "exchange": "factorised, direct reduce-scatter + all-gather on every peer link"},
            "all_gather_ms": t_gather, "all_reduce_ms": t_reduce, "predicted_exposed_exchange_ms": exposed,
            "measured_per_rank_compute_ms_max": compute, "predicted_ms_per_step": step,
            "predicted_views_per_s": world * views_per_rank / (step * 1e-3),
            "measured_ms_per_step": measured_ms_per_step, "measured_over_predicted": measured_ms_per_step / step,
            "product_iteration_model": {
                "note": "libgstrain's train_step = this raster step + the optimizer (fused Adam, 0.28 ms per 10^6 splats); NOT what this line measures. "
                        "unpipelined: one geometry all-reduce behind A9, SH rebuild + SH Adam under it; pipelined (DVS_EXCHANGE_PIPELINE=1, 4 chunks): A9 "
                        "chunk -> all-reduce chunk -> geometry Adam chunk -> next iteration's A2 chunk. A model until a multi-GPU run exists.",
                "optimizer_ms": optimizer_ms, "unpipelined_exposed_ms": exposed_u, "unpipelined_ms_per_iteration": step_u,
                "unpipelined_views_per_s": world * views_per_rank / (step_u * 1e-3), "pipelined_chunks": k, "pipelined_exposed_ms": exposed_p,
                "pipelined_ms_per_iteration": step_p, "pipelined_views_per_s": world * views_per_rank / (step_p * 1e-3)}}


def watchdog_seconds():
    try:
        return max(30.0, float(os.environ.get("DVS_BENCH_WATCHDOG_S", "900")))
    except ValueError:
        return 900.0


def watchdog_record(args, message):
    """The one JSON line of a run that did not finish: same keys as a result, value null, the reason in "error"."""
    return {"metric": "train views/sec (fwd+bwd raster)", "value": None, "unit": "views/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic", "config": {"workload": args.workload}, "error": message}


def arm_rank_watchdog(args, rank):
    """Inside a rank (also under an external launcher): a daemon timer that ends the process with an error line on rank 0 if the run has
    not finished in time — a hung RCCL rendezvous or collective otherwise blocks in C++ where no Python exception can reach it."""
    import threading

    def fire():
        if rank == 0:
            print(json.dumps(watchdog_record(args, f"rank 0 still running after {watchdog_seconds():.0f} s (hung collective or rendezvous?): aborted by bench.py's watchdog")), flush=True)
        sys.stderr.write(f"[bench.py] rank {rank}: watchdog expired, exiting\n"); sys.stderr.flush()
        os._exit(3)
    t = threading.Timer(watchdog_seconds() + (0.0 if rank == 0 else 10.0), fire)      # rank 0 first: its line is the one the caller reads
    t.daemon = True
    t.start()
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)       # 20 warm-up + 200 timed iterations
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="C3", choices=list(WORKLOADS))
    ap.add_argument("--absgrad", type=int, default=1, help="accumulate |dL/dmean2D| (reference default --absgrad true, main.cpp:44)")
    ap.add_argument("--mode", default="batch", choices=["batch", "grouped", "pipelined"],
                    help="batch (default, the measured winner): the views of a rank go through ONE multi-view pass (dvs_raster_forward_views: "
                         "parameters read once, one sort / scan / composite launch per step); grouped: --groups passes pipelined over two "
                         "contexts / streams; pipelined: one view per pass (the round-1 shape)")
    ap.add_argument("--groups", type=int, default=2, help="grouped mode: multi-view groups per rank and step")
    ap.add_argument("--stagger", type=int, default=0,
                    help="grouped / pipelined: 1 = a group's forward starts only when the previous group's forward has been composited, so that "
                         "its HBM-bound front end runs under the previous group's VALU-bound backward instead of beside its front end")
    ap.add_argument("--contexts", type=int, default=2, help="rasterizer contexts / HIP streams the groups of a step are pipelined over")
    ap.add_argument("--exchange", default="auto", choices=["auto", "factorised", "allreduce"],
                    help="N>1 gradient exchange: one all-reduce of all 236 B/splat, or factorised (all-reduce of 44 B + all-gather of "
                         "12 B per splat per view, SH rows rebuilt locally, gathers overlapped with compute); auto = factorised")
    ap.add_argument("--exchange-impl", default="dvs_comm", choices=["dvs_comm", "torch"],
                    help="N>1: who runs the collectives. dvs_comm (default) = the product's communication layer, include/dvs_comm.h (librccl "
                         "behind plain C, the calls libgstrain.so's train_step makes), with its stream / event choreography: early colour "
                         "all-gather, optionally chunked A9 with one grouped launch per chunk, SH rebuild under the geometry all-reduce; no "
                         "torch.distributed at all (barrier and max-over-ranks go through the same communicator). torch = "
                         "torch.distributed (backend nccl = RCCL), the path of rounds 1-4, kept for A/B")
    ap.add_argument("--global-views", type=int, default=8, help="views per training iteration, sharded over the GPUs (BASELINE config C4: 8)")
    ap.add_argument("--views-per-step", type=int, default=0,
                    help="if > 0: this many views PER GPU per step instead of sharding --global-views (weak scaling, the round-1 shape)")
    ap.add_argument("--shn-tiled", type=int, default=1,
                    help="1: shN parameters/gradients in the DVS_SHN_TILED HBM layout (default); 0: the reference's [N,45] rows")
    ap.add_argument("--bwd-variant", default="tr", choices=["tr", "blocks", "reduce", "mm"],
                    help="A8 kernel (dvs_set_backward_variant): tr = default (round 3, measured winner); blocks (round 2) / reduce (round 1) / mm = the measured alternatives")
    ap.add_argument("--fwd-variant", default="quadrant", choices=["blocks", "quadrant"], help="A7 kernel (dvs_set_forward_variant)")
    ap.add_argument("--grad-mode", type=int, default=0, help="dvs_opts.grad_mode: 0 = DVS_GRAD_TRUE, 1 = DVS_GRAD_LINEAGE (same cost)")
    ap.add_argument("--early-gather", type=int, default=1,
                    help="N>1, factorised exchange, --mode batch: 1 = the colour gradients are taken from the composite backward's rows "
                         "(dvs_raster_backward_dcolor) and their all-gather runs under the preprocess backward (A9)")
    ap.add_argument("--a9-chunks", type=int, default=1,
                    help="N>1, factorised exchange with the early gather: A9 runs in this many splat chunks and each chunk's 44 B/splat geometry "
                         "all-reduce starts on the side stream as soon as its A9 launch is queued (SURVEY.md 8(e)); 1 (default) = one all-reduce "
                         "at the end. Measured on the one-view-per-rank step over a 1-rank RCCL communicator: every chunk costs this Python "
                         "host 0.06-0.1 ms of launches (four collectives + an event each; 1.02 / 1.14 / 1.23 ms per step at 1 / 2 / 4 chunks) and "
                         "the step turns host-bound — libgstrain.so issues a chunk as one ncclGroup from C++ (7 us per chunk) and defaults to 4")
    ap.add_argument("--graph", type=int, default=0,
                    help="1: capture the step (one multi-view pass forward + loss gradient + backward) into a HIP graph after the warm-up and "
                         "replay it (one GPU, --mode batch, asynchronous forward: the pass has no host synchronisation and fixed launch shapes)")
    ap.add_argument("--async-forward", type=int, default=1, help="1: dvs_set_async — the forward never synchronises the host (T stays on the device)")
    ap.add_argument("--tight-tiles", type=int, default=0,
                    help="1: dvs_opts.tile_bounds = DVS_TILES_TIGHT (opt-in: only the tiles the alpha >= 1/255 ellipse reaches; same images and "
                         "gradients, shorter lists). The headline (default 0) runs the canonical 3-sigma rectangles")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-iters", type=int, default=10, help="extra iterations with per-stage hipEvent timing")
    args = ap.parse_args()

    # `python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves (one process per GPU under
    # torch.distributed.run on 127.0.0.1, a free port); rank 0's JSON line passes through. Under a launcher (WORLD_SIZE set) run as a rank.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), MASTER_ADDR="127.0.0.1")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        # watchdog: a rank stuck in a rendezvous or a collective must not hold the caller until ITS limit — after DVS_BENCH_WATCHDOG_S
        # (default 900 s) the launcher's process group (started here, its own session) is killed and an error line is printed instead
        child = subprocess.Popen(cmd, env=env, start_new_session=True)
        try:
            raise SystemExit(child.wait(timeout=watchdog_seconds() + 20.0))        # (the ranks' own watchdogs fire first and say more)
        except subprocess.TimeoutExpired:
            import signal
            try:
                os.killpg(child.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            print(json.dumps(watchdog_record(args, f"no result within {watchdog_seconds():.0f} s: the {args.gpus}-rank run was killed by bench.py's watchdog")), flush=True)
            raise SystemExit(3)

    import numpy as np
    import torch
    import divshot_amd as dv
    from divshot_amd.raster import Rasterizer, params_to_device
    from divshot_amd.parallel import GradBuffer, FactorisedExchange

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE)")
    watchdog = arm_rank_watchdog(args, rank) if world > 1 else None
    if os.environ.get("DVS_BENCH_TEST_HANG") == "1" and world > 1:      # tests/test_parallel.py: a rank that never comes back
        time.sleep(1e6)
    dist = None
    ndev = max(1, torch.cuda.device_count())
    if world > ndev and args.exchange_impl == "torch" and os.environ.get("DVS_DIST_BACKEND", "nccl") == "nccl":
        raise SystemExit(f"bench.py: {world} ranks need {world} GPUs over RCCL, this node has {ndev} "
                         "(DVS_DIST_BACKEND=gloo lets a functional test oversubscribe one GPU)")
    dev_index = local_rank % ndev          # (a functional test may oversubscribe one GPU with a gloo group; normally 1 rank = 1 GPU)
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    multi = world > 1 or os.environ.get("DVS_FORCE_COLLECTIVES") == "1"     # (forced: the N>1 step over a 1-rank communicator — a hardware test of the path)
    comm = None
    if multi and args.exchange_impl == "torch":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        backend = os.environ.get("DVS_DIST_BACKEND", "nccl")       # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=watchdog_seconds()))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=watchdog_seconds()))
    elif multi:
        # the product's layer: librccl through include/dvs_comm.h; rank 0 serves the RCCL id on MASTER_PORT + 1789 (DVS_COMM_PORT).
        # DVS_COMM_BACKEND=tcp (tests only: two ranks on one GPU) is announced on stderr and recorded in the line.
        from divshot_amd.parallel import DvsComm
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if world > ndev and os.environ.get("DVS_COMM_BACKEND") != "tcp":
            raise SystemExit(f"bench.py: {world} ranks need {world} GPUs over RCCL, this node has {ndev} (DVS_COMM_BACKEND=tcp lets a functional test share one GPU)")
        comm = DvsComm(dev_index, rank, world)

    def barrier():
        if comm is not None:
            comm.barrier()
        elif dist is not None:
            dist.barrier()

    def max_over_ranks(x):
        if comm is not None:
            return comm.max_over_ranks(x)
        if dist is not None:
            t_ = torch.tensor([x], dtype=torch.float64, device=dev)
            dist.all_reduce(t_, op=dist.ReduceOp.MAX)
            return float(t_.item())
        return x

    n, W, H, deg, soff = WORKLOADS[args.workload]
    weak = args.views_per_step > 0
    if weak:                                      # V views per GPU per step
        VPS = args.views_per_step
        n_cams = max(8, world * VPS)
        my_views = [(rank * VPS + v) % n_cams for v in range(VPS)]
        views_of = lambda r: [(r * VPS + v) % n_cams for v in range(VPS)]
    else:                                         # config C4: one iteration = --global-views views, view g on rank g % world
        if args.global_views % world != 0:
            raise SystemExit(f"bench.py: --global-views {args.global_views} must be a multiple of the number of GPUs ({world})")
        VPS = args.global_views // world
        n_cams = max(8, args.global_views)
        views_of = lambda r: list(range(r, args.global_views, world))
        my_views = views_of(rank)
    GLOBAL_VIEWS = world * VPS                    # views per step over all ranks
    spec = dv.make_spec(n, W, H, sh_degree=deg, n_cams=n_cams, scale_log_offset=soff)
    P = dv.synth_splats(spec)                     # identical replica on every rank (same seed)
    cams = [dv.synth_camera(spec, i) for i in my_views]
    targets = [torch.from_numpy(dv.synth_target(spec, i)).to(dev) for i in my_views]
    cam, target = cams[0], targets[0]
    params = params_to_device(P, dev)
    # The rank's views of a step form K groups; a group goes through the multi-view pass of the C-ABI (dvs_raster_forward_views /
    # dvs_raster_backward_*: parameters read once, one depth sort / scan / (view, tile) sort / composite launch for the group), and
    # the groups are software-pipelined over two rasterizer contexts on two HIP streams: the HBM-bound front of group g+1 (A2, sorts)
    # runs under the VALU-bound composite kernels of group g, the A9 of group g under the composite backward of group g+1.
    #   --mode batch = 1 group (everything in lockstep) | grouped = --groups K (default 2) | pipelined = one view per group (round 1)
    K = 1 if args.mode == "batch" else (VPS if args.mode == "pipelined" else max(1, min(args.groups, VPS)))
    while VPS % K:
        K -= 1
    G = VPS // K                                  # views per group
    n_ctx = max(1, min(args.contexts, K))
    rasts = [Rasterizer(dev_index, max_splats=n, max_w=W, max_h=H, max_views=G) for _ in range(n_ctx)]
    for r_ in rasts:
        r_.set_backward_variant(args.bwd_variant)
        r_.set_forward_variant(args.fwd_variant)
        r_.set_async(bool(args.async_forward))
    rast = rasts[0]
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_ctx)] if n_ctx > 1 else [torch.cuda.current_stream(dev)]
    tiled = bool(args.shn_tiled)
    if tiled:       # training-loop layout of the 45 higher-order SH floats (DVS_SHN_TILED); converted once, outside the timed region
        params["shN"] = rast.shn_relayout(params["shN"], n, to_tiled=True)
    # one flat gradient buffer so the exchange is a single large collective; the views of a step accumulate into it
    gbuf = GradBuffer(n, dev, shn_tiled=tiled)
    flat, grads = gbuf.flat, dict(gbuf.views)
    if args.absgrad:
        grads["absgrad2d"] = torch.zeros((n, 2), dtype=torch.float32, device=dev)
    outs = [torch.empty((G, 3, H, W), dtype=torch.float32, device=dev) for _ in range(n_ctx)]
    out = torch.empty((3, H, W), dtype=torch.float32, device=dev)
    inv_P = 1.0 / (W * H)
    # upstream gradient dL/drgb = (rgb - target) / P formed by ONE elementwise kernel per group: rgb * (1/P) + (-target / P), the
    # second term prepared once (the targets are constant inputs)
    neg_targets_scaled = [(-t_ * inv_P).contiguous() for t_ in targets]
    neg_targets_group = [torch.stack(neg_targets_scaled[gi * G:(gi + 1) * G]).contiguous() for gi in range(K)]
    cams_group = [cams[gi * G:(gi + 1) * G] for gi in range(K)]

    exchange = args.exchange
    if exchange == "auto":
        # exposed bytes received per splat per rank: ring all-reduce of B bytes ~ 2 (w-1)/w B; all-gather of b bytes ~ (w-1) b.
        # The factorised exchange gathers the views of all but the last group under the compute of the following group:
        # (w-1) 12 G + ring 44 < ring 236 for every world size.
        exchange = "factorised"
    factorised = multi and exchange == "factorised"
    early = factorised and bool(args.early_gather) and K == 1
    if comm is not None and factorised and not early:
        raise SystemExit("bench.py: --exchange-impl dvs_comm runs the product's choreography (--mode batch with the early gather); use --exchange-impl torch for the other modes")
    if factorised:
        if comm is not None:
            from divshot_amd.parallel import DvsCommExchange
            fx = DvsCommExchange(n, dev, comm, views_per_rank=VPS)
        else:
            fx = FactorisedExchange(n, dev, world, views_per_rank=VPS, rank_major=early)
        dcolor_scratch = torch.empty((VPS, n, 3), dtype=torch.float32, device=dev) if early else None
        campos_all = np.array([list(dv.synth_camera(spec, views_of(r)[v]).campos) for r, v in fx.slots()], np.float32)
        # rebuild the SH rows of a view's slots right behind its all-gather, on the exchange's side stream
        if not early:
            fx.set_combiner(lambda lo, hi, acc: rast.sh_grad_combine(params["pos"], campos_all[lo:hi], fx.dcolor_all[lo:hi], gbuf.views["sh0"],
                                                                     gbuf.views["shN"], deg, accumulate=acc, shn_tiled=tiled))
    a9_chunks = None
    if early and args.a9_chunks > 1 and tiled:
        per = ((n + args.a9_chunks - 1) // args.a9_chunks + 255) // 256 * 256
        a9_chunks = [(f, min(per, n - f)) for f in range(0, n, per)]
    chunk_done = [torch.cuda.Event() for _ in (a9_chunks or [])]
    dcol_done = torch.cuda.Event()
    bwd_done = [torch.cuda.Event() for _ in range(n_ctx)]
    fwd_done = [torch.cuda.Event() for _ in range(n_ctx)]
    step_done = torch.cuda.Event()
    main_stream = torch.cuda.current_stream(dev)
    comm_marks = []                               # (event before the exposed exchange, event after it) per timed step
    step_marks = []                               # (event at the start of a timed step, event before its exchange): this rank's compute
    torch.cuda.synchronize()

    gm = {"mode": args.grad_mode}            # (mutable: the lineage-mode side measurement below re-times the same step in grad_mode 1)

    def step(timed=False):
        e_start = None
        if timed and multi:
            e_start = torch.cuda.Event(enable_timing=True); e_start.record(main_stream)
        if n_ctx > 1:
            step_done.record(main_stream)          # everything enqueued so far (previous step incl. its exchange)
        for gi in range(K):
            c = gi % n_ctx
            st = streams[c] if n_ctx > 1 else torch.cuda.current_stream(dev)        # (one context: whatever stream is current — also a capturing one)
            with torch.cuda.stream(st):
                if n_ctx > 1 and gi < n_ctx:
                    st.wait_event(step_done)       # no overlap across steps: the next step's views see updated parameters
                if n_ctx > 1 and args.stagger and gi > 0:
                    st.wait_event(fwd_done[(gi - 1) % n_ctx])
                imgs = rasts[c].forward_views(params, cams_group[gi], sh_degree=deg, absgrad=bool(args.absgrad), out=outs[c], shn_tiled=tiled, tight_tiles=bool(args.tight_tiles),
                                              grad_mode=gm["mode"])
                if n_ctx > 1:
                    fwd_done[c].record(st)
                dL = torch.add(neg_targets_group[gi], imgs, alpha=inv_P)
                g = grads
                if factorised:
                    g = dict(grads); g["dcolor"] = dcolor_scratch if early else fx.dcolor_local[gi * G:(gi + 1) * G]
                # A8 (composite backward) writes only this context's intermediate rows: it needs no ordering against the other
                # group. Only A9, which accumulates into the shared gradient rows non-atomically, runs in group order.
                rasts[c].backward_composite(dL)
                if early:          # the colour gradients leave before A9: their all-gather runs under it, on the exchange's side stream
                    rasts[c].backward_dcolor(fx.dcolor_local)
                    dcol_done.record(st)
                    fx.gather_all(dcol_done)
                if n_ctx > 1 and gi > 0:
                    st.wait_event(bwd_done[(gi - 1) % n_ctx])
                if early and a9_chunks:           # A9 chunk by chunk, each chunk's geometry all-reduce leaving behind it on the side stream
                    def _after(k_, first_, count_):
                        chunk_done[k_].record(st)
                        fx.reduce_geometry_chunk(gbuf, first_, count_, chunk_done[k_])
                    rasts[c].backward_project_chunks(g, a9_chunks, _after, accumulate=(gi > 0))
                else:
                    rasts[c].backward_project(grads=g, accumulate=(gi > 0), factorised_sh=factorised)
                if n_ctx > 1:
                    bwd_done[c].record(st)
                if factorised and gi < K - 1:
                    for v in range(gi * G, (gi + 1) * G):
                        fx.gather_view(v, bwd_done[c] if n_ctx > 1 else None)      # overlaps with the next group's kernels
        if n_ctx > 1:
            main_stream.wait_event(bwd_done[(K - 1) % n_ctx])
        if multi:
            ea = None
            if timed:
                ea = torch.cuda.Event(enable_timing=True); ea.record(main_stream)
            if factorised:
                fx.exchange(gbuf, rast, params["pos"], campos_all, deg, shn_tiled=tiled)
            elif comm is not None:
                comm.all_reduce_sum(flat)
            else:
                dist.all_reduce(flat)
            if timed:
                eb = torch.cuda.Event(enable_timing=True); eb.record(main_stream)
                comm_marks.append((ea, eb))
                step_marks.append((e_start, ea))

    # (no fallback: if the exchange cannot run on this stack the bench fails loudly instead of measuring something else)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    run_step = step
    graph = None
    if args.graph:
        if multi or n_ctx > 1 or not args.async_forward:
            raise SystemExit("bench.py: --graph needs one GPU, one context (--mode batch) and --async-forward 1")
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        run_step = lambda timed=False: graph.replay()
        for _ in range(3):
            run_step()
        torch.cuda.synchronize()
    # one timing event per step boundary on the main stream (recorded, never waited on inside the loop): steps do not overlap, so
    # the deltas are the per-step durations; read after the timed region for the p10 / median / p90 spread
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    for i_ in range(args.steps):
        marks[i_].record(main_stream)
        run_step(timed=True)
    marks[args.steps].record(main_stream)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed)

    # the same step in the other gradient mode (libgstrain.so runs DVS_GRAD_LINEAGE, the headline DVS_GRAD_TRUE): a short timed
    # side loop, so that "same cost" is a measured statement
    other_mode = None
    if graph is None and args.steps >= 10:
        gm["mode"] = 1 - args.grad_mode
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        barrier()
        k_other = max(5, min(50, args.steps))
        t1 = time.perf_counter()
        for _ in range(k_other):
            step()
        torch.cuda.synchronize()
        barrier()
        el = max_over_ranks(time.perf_counter() - t1)
        other_mode = {"grad_mode": gm["mode"], "steps": k_other, "views_per_s": GLOBAL_VIEWS * k_other / el, "ms_per_step": el / k_other * 1e3}
        gm["mode"] = args.grad_mode
        step(); torch.cuda.synchronize()          # leave the gradient buffer as the timed region left it (norms below)
    for r_ in rasts:
        r_.get_num_rendered()         # raises if an asynchronous forward of the timed region overflowed its instance arena
    grad_norms = {k: float(v.double().norm()) for k, v in gbuf.views.items()}      # after the exchange: identical on every rank
                                                                                   # (taken before the profiling iterations reuse the buffer)
    # rccl-tests-style microbenchmark of the collectives the exchange is made of, at the exchange's sizes; outside the
    # timed region, on scratch buffers. busbw uses the rccl-tests convention (all-reduce 2(N-1)/N, all-gather (N-1)/N of the total).
    comm_micro = None
    per_rank_compute_ms = None
    if multi:
        def _coll_ms(fn, iters=10):
            for _ in range(3):
                fn()
            torch.cuda.synchronize(); barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record(); torch.cuda.synchronize()
            return max_over_ranks(e0.elapsed_time(e1) / iters * 1e-3) * 1e3
        s_geom = torch.zeros_like(gbuf.flat_geom); s_flat = torch.zeros_like(gbuf.flat)
        s_loc = torch.zeros((VPS * n * 3,), device=dev); s_all = torch.zeros((world * VPS * n * 3,), device=dev)
        if comm is not None:
            f_geom, f_flat, f_gather = (lambda: comm.all_reduce_sum(s_geom)), (lambda: comm.all_reduce_sum(s_flat)), (lambda: comm.all_gather(s_loc, s_all))
        else:
            f_geom, f_flat, f_gather = (lambda: dist.all_reduce(s_geom)), (lambda: dist.all_reduce(s_flat)), (lambda: dist.all_gather_into_tensor(s_all, s_loc))
        comm_micro = {}
        for name, nbytes, fac, fn in (
                ("all_reduce_geometry_44B_per_splat", s_geom.numel() * 4, 2.0 * (world - 1) / world, f_geom),
                ("all_reduce_full_rows_236B_per_splat", s_flat.numel() * 4, 2.0 * (world - 1) / world, f_flat),
                ("all_gather_dcolor_12B_per_splat_and_view", s_all.numel() * 4, (world - 1) / world, f_gather)):
            ms_ = _coll_ms(fn)
            comm_micro[name] = {"bytes": nbytes, "ms": ms_, "algbw_GBps": nbytes / ms_ / 1e6, "busbw_GBps": fac * nbytes / ms_ / 1e6}
        del s_geom, s_flat, s_loc, s_all
        # this rank's compute per step = start of the step -> just before the exposed exchange (the early colour all-gather and, when
        # chunked, the geometry groups already run underneath on the communication stream); gathered from every rank
        if step_marks:
            torch.cuda.synchronize()
            mine = sum(a.elapsed_time(b) for a, b in step_marks) / len(step_marks)
            if comm is not None:
                per_rank_compute_ms = comm.gather_floats(mine)
            else:
                tl = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
                dist.all_gather(tl, torch.tensor([mine], dtype=torch.float64, device=dev))
                per_rank_compute_ms = [float(t_.item()) for t_ in tl]
    # everything below measures ONE view at a time: its own single-view context (the step's contexts are sized and primed for groups)
    rast1 = Rasterizer(dev_index, max_splats=n, max_w=W, max_h=H)
    rast1.set_backward_variant(args.bwd_variant); rast1.set_forward_variant(args.fwd_variant); rast1.set_async(bool(args.async_forward))
    if rank == 0:
        rast1.forward(params, cam, sh_degree=deg, absgrad=bool(args.absgrad), out=out, shn_tiled=tiled, grad_mode=args.grad_mode, tight_tiles=bool(args.tight_tiles))
    # ---- strict single-view figure of SURVEY.md §8(d): 1 / (t_fwd + t_bwd), one view at a time on one stream, no pipelining ----
    strict = None
    if rank == 0 and args.profile_iters > 0:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n_strict = max(10, min(100, args.steps))
        g_strict = {k: v for k, v in grads.items() if k != "dcolor"}
        for it_ in range(n_strict + 5):
            if it_ == 5:
                ev0.record(main_stream)
            img = rast1.forward(params, cam, sh_degree=deg, absgrad=bool(args.absgrad), out=out, shn_tiled=tiled, grad_mode=args.grad_mode, tight_tiles=bool(args.tight_tiles))
            dL = torch.add(neg_targets_scaled[0], img, alpha=inv_P)
            rast1.backward(dL, grads=g_strict)
        ev1.record(main_stream)
        torch.cuda.synchronize()
        ms_view = ev0.elapsed_time(ev1) / n_strict
        strict = {"ms_per_view": ms_view, "views_per_s": 1e3 / ms_view, "views": n_strict,
                  "note": "one view at a time on one stream (forward" + ("" if args.async_forward else " incl. its host sync on T") + ", upstream gradient, backward); hipEvents",
                  "async_forward": bool(args.async_forward)}
    clocks = read_clocks(dev_index) if rank == 0 else None
    # ---- the composite kernels timed inside the real step: a replica of the timed region with hipEvent pairs around k_render_fwd /
    # k_render_bwd on the streams they are launched on, never synchronised in between (dvs_enable_kernel_probe). Same concurrency as
    # the timed region (views pipelined over two streams), so these are the durations rocprofv3 sees for the same command.
    probe = {}
    if args.profile_iters > 0:
        for r_ in rasts:
            r_.kernel_probe(True)
        for _ in range(max(1, min(args.steps, 10))):
            step()
        torch.cuda.synchronize()
        acc_p = {"render_fwd": [0.0, 0], "render_bwd": [0.0, 0]}
        for r_ in rasts:
            for k_, (ms_, cnt_) in r_.read_kernel_probe().items():
                acc_p[k_][0] += ms_ * cnt_; acc_p[k_][1] += cnt_
            r_.kernel_probe(False)
        probe = {k_: (v_[0] / v_[1] if v_[1] else None) for k_, v_ in acc_p.items()}
        barrier()
    # ---- per-stage hipEvent timing of the STEP's own multi-view pass (one GPU, batch mode): the launches the timed region makes, with
    # stage timing on (each call then synchronises; kernel durations are unaffected) — feeds roofline.kernels[]
    batch_stage_ms = {}
    if rank == 0 and args.profile_iters > 0 and world == 1 and K == 1 and not args.graph:
        rasts[0].enable_timing(True)
        acc_b = {}
        for _ in range(max(2, min(args.profile_iters, 5))):
            step(); torch.cuda.synchronize()
            for k, v in rasts[0].stage_timing().items():
                acc_b.setdefault(k, []).append(v)
        rasts[0].enable_timing(False)
        batch_stage_ms = {k: float(np.mean(v)) for k, v in acc_b.items()}
    # ---- per-stage hipEvent timing (separate iterations; timing mode synchronises per call) --------------
    stage_ms = {}
    if rank == 0 and args.profile_iters > 0:
        rast1.enable_timing(True)
        acc = {}
        for _ in range(args.profile_iters):
            img = rast1.forward(params, cam, sh_degree=deg, absgrad=bool(args.absgrad), out=out, shn_tiled=tiled, tight_tiles=bool(args.tight_tiles))
            dL = (img - target) * inv_P
            rast1.backward(dL, grads={k: v for k, v in grads.items() if k != "dcolor"})
            for k, v in rast1.stage_timing().items():
                acc.setdefault(k, []).append(v)
        rast1.enable_timing(False)
        stage_ms = {k: float(np.mean(v)) for k, v in acc.items()}
    barrier()

    exchange_info = None
    if multi:
        exchange_info = {
            "impl": args.exchange_impl + (": include/dvs_comm.h, the calls of libgstrain.so's train_step" if comm is not None else ": torch.distributed"),
            "backend": comm.backend if comm is not None else ("torch.distributed/" + dist.get_backend()),
            # what the COMMUNICATOR itself reports (ncclCommCount through dvs_comm_backend_ranks), not the launcher's WORLD_SIZE
            "rccl_nranks": comm.backend_ranks if comm is not None else dist.get_world_size(),
            "world_size_env": world, "exchange": exchange, "early_gather": bool(early), "a9_chunks": len(a9_chunks) if a9_chunks else 1,
            "choreography": ("colour all-gather on the communication stream behind dvs_raster_backward_dcolor (under A9); "
                             + ("A9 in %d splat chunks, each chunk's geometry groups as ONE grouped launch behind it; " % len(a9_chunks) if a9_chunks else "geometry all-reduce behind A9; ")
                             + "SH rows rebuilt on the compute stream when the gather has landed, under the geometry all-reduce") if factorised else "one all-reduce of the flat gradient buffer",
        }
    if rank == 0:
        st = rast1.state
        V = int((torch.from_numpy(rast1._d2h(st.radii, (n,), np.int32)) > 0).sum())
        T = int(rast1.get_num_rendered())          # (synchronises; also raises on an instance-arena overflow during the run)
        Ppix = W * H
        tiles = st.tiles_x * st.tiles_y
        ab, p = algorithmic_bytes(n, V, T, Ppix, tiles, deg, bool(args.absgrad))
        total_bytes = sum(ab.values())
        ms_per_step = elapsed / args.steps * 1e3
        try:
            pr = torch.cuda.get_device_properties(dev)
            device_info = {"name": pr.name, "gcn_arch": getattr(pr, "gcnArchName", None), "compute_units": pr.multi_processor_count,
                           "clock_mhz": getattr(pr, "clock_rate", 0) / 1e3 if getattr(pr, "clock_rate", 0) else None,
                           "memory_gb": round(pr.total_memory / 2 ** 30, 1), "host_cores": os.cpu_count()}
        except Exception:      # noqa: BLE001
            device_info = None
        step_ms = sorted(marks[i_].elapsed_time(marks[i_ + 1]) for i_ in range(args.steps))
        step_spread = [step_ms[int(q * (len(step_ms) - 1))] for q in (0.1, 0.5, 0.9)] if step_ms else None
        value = GLOBAL_VIEWS * args.steps / elapsed
        comm_ms = None
        if comm_marks:
            cm = sorted(a_.elapsed_time(b_) for a_, b_ in comm_marks)
            comm_ms = {"mean": float(np.mean(cm)), "p50": cm[len(cm) // 2], "p90": cm[int(0.9 * (len(cm) - 1))],
                       "note": "rank 0, main stream: from 'last view's backward done' to 'exchange complete' (the gathers of earlier views "
                               "run under compute and are not in this figure)"}
        # dominant kernel = the longest single-kernel stage
        single = {k: stage_ms[k] for k in ("render_bwd", "render_fwd", "preprocess_fwd", "preprocess_bwd", "duplicate") if k in stage_ms}
        roofline = None
        if single:
            dom = max(single, key=single.get)
            in_step_ms = probe.get(dom)                       # mean launch duration inside the pipelined step (kernel probe)
            dur_ms = in_step_ms if in_step_ms else single[dom]
            views_per_launch = G if in_step_ms else 1          # a launch inside the step composites all views of its group
            achieved = ab[dom] * views_per_launch / (dur_ms * 1e-3) / 1e9
            kern = "k_" + dom
            if dom == "render_bwd":
                kern = {"blocks": "k_render_bwd_blocks<", "reduce": "k_render_bwd<", "mm": "k_render_bwd_mm<", "tr": "k_render_bwd_tr<"}[args.bwd_variant]
            traffic, traffic_src, same_run, counter_bytes_step, counters_prov = None, None, False, None, None
            try:        # HBM bytes per launch from the committed PMC passes (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 runs)
                import glob
                tfile = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))[-1]      # the latest round's PMC passes
                tj = json.load(open(tfile))
                # the counters belong to the workload and batch shape they were collected on; any other run reports null
                same_run = args.workload == tj.get("workload", "C3") and views_per_launch == tj.get("views_per_launch", 8)
                if same_run and world == 1:
                    counter_bytes_step = tj.get("counter_bytes_per_step")
                for kname, rec_ in tj["kernels"].items():
                    if same_run and kname.startswith(kern):
                        traffic = rec_["hbm_bytes_per_launch_corrected"]
                        traffic_src = "profiles/" + os.path.basename(tfile) + " (2*FETCH_SIZE + WRITE_SIZE, KB->B)"
                counters_prov = counters_provenance(tfile, tj)
            except Exception:
                pass
            # VALU issue occupancy of the same kernel from the committed SQ counter pass (its own rocprofv3 run): SQ_ACTIVE_INST_VALU
            # counts 4-cycle quads per SIMD, so quads * 4 / (SIMDs * duration * clock) is the fraction of the kernel's duration the
            # vector ALUs were issuing. It is ~1.0 for k_render_bwd: the kernel is VALU-bound, which is why its HBM fraction is low.
            valu = None
            try:
                import glob, re
                sq = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_sq.txt")))[-1]
                dur_us = act = insts = gui = None
                for line in open(sq):
                    if not line.lstrip().startswith(kern):
                        continue
                    f = line.split()
                    if "SQ_ACTIVE_INST_VALU" in f:
                        act = float(f[-1])
                    elif "SQ_INSTS_VALU" in f:
                        insts = float(f[-1])
                    elif "GRBM_GUI_ACTIVE" in f:
                        gui = float(f[-1])
                    elif dur_us is None and len(f) > 6 and re.fullmatch(r"[0-9.]+", f[-10] or ""):
                        dur_us = float(f[-10])          # avg_us column of the kernel-trace table
                if dur_us and act and same_run:
                    valu = {"source": "profiles/" + os.path.basename(sq), "insts_valu_per_launch": insts, "active_quads_per_launch": act,
                            "avg_us_under_pmc": dur_us}
                    if gui:
                        # GRBM_GUI_ACTIVE = cycles the GPU was busy during the dispatch (summed over the 8 XCDs when it exceeds what one
                        # clock domain can tick in the duration): the measured shader clock, and the VALU busy fraction against it
                        units = 8 if gui / (dur_us * 1e-6) > 3.0e9 else 1
                        clk = gui / units / (dur_us * 1e-6)
                        raw = act * 4.0 / (1024 * (gui / units))
                        valu.update({"GRBM_GUI_ACTIVE_per_launch": gui, "measured_clock_GHz": clk / 1e9,
                                     # SQ_ACTIVE_INST_VALU in 4-cycle quads over all SIMDs against SIMDs x busy cycles. The two counters come
                                     # from different blocks (SQ per SIMD, GRBM per XCD) and the ratio lands a few per cent above 1 when the
                                     # vector ALUs never idle: reported as a fraction capped at 1, with the raw ratio beside it
                                     "active_quads_x4_over_simd_cycles_raw": raw,                 # (the capped "busy fraction" of rounds 4-5 is gone: see roofline.secondary)
                                     "note": "an upper bound on how busy the vector ALUs are, not proof that issue is what limits the kernel: the "
                                             "round-4 ablations (profiles/r04_a8_ablation.txt) show the per-batch latency chain (staging gather, "
                                             "table read-add-write passes) costs more than the arithmetic"})
                    else:
                        valu["busy_fraction_note"] = "no GRBM_GUI_ACTIVE in this counter pass: the clock during the kernel is unknown, no busy fraction is derived"
            except Exception:
                pass
            roofline = {"bound": "hbm", "kernel": kern.rstrip("<"), "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src, "counters": counters_prov,
                        "algorithmic_bytes_per_launch": ab[dom] * views_per_launch, "views_per_launch": views_per_launch, "avg_launch_ms": dur_ms,
                        "avg_launch_ms_isolated": single[dom], "avg_launch_source": ("hipEvent pairs around the kernel inside a replica of the timed "
                        "region (dvs_enable_kernel_probe)" if in_step_ms else "per-stage hipEvent timing, one view at a time"), "valu_issue": valu,
                        "note": "k_render_bwd is VALU-issue-bound (SQ_ACTIVE_INST_VALU ~ kernel duration, profiles/r*_pmc_sq.txt), "
                                "so its HBM fraction is low by construction; see DESIGN.md section 5"}
            # the ceiling the composite kernels are actually against (SURVEY 8(d) "secondary ceilings"): vector-instruction
            # issue = sum over instruction classes of (dynamic count x measured issue cost) / (kernel cycles x SIMDs), tools/valu_ceiling.py
            try:
                import glob
                vfile = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_valu_ceiling.json")))[-1]
                vj = json.load(open(vfile))
                sec = {}
                for kn_, rec_ in vj["kernels"].items():
                    sec[kn_] = {"frac": rec_["frac_of_valu_issue_ceiling"], "valu_wave_instructions_per_launch": rec_["valu_wave_instructions_per_launch"],
                                "issue_cycles_per_launch": rec_["issue_cycles_per_launch"], "kernel_cycles_per_launch": rec_["kernel_cycles_per_launch"],
                                "clock_GHz": rec_["clock_GHz"]}
                dom_key = next((k_ for k_ in sec if k_.startswith(kern.rstrip("<"))), None)
                roofline["secondary"] = {"bound": "valu_issue", "frac": sec[dom_key]["frac"] if dom_key else None, "kernel": dom_key, "kernels": sec,
                                         "source": "profiles/" + os.path.basename(vfile),
                                         "definition": "sum over instruction classes (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32, _INT32, _CVT, rest) of count per launch x issue "
                                                       "cost in SIMD cycles (tools/ubench/valu_cost at its measured clock) / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs); the "
                                                       "'rest' class is priced with the kernel's static mix; accounting uncertainty ~ +-10 %: a value near 1 means the "
                                                       "kernel is at the vector-issue ceiling of its instruction mix",
                                         "collected_on_these_kernels": vj.get("kernel_source_sha16") == kernel_source_sha()}
            except Exception:      # noqa: BLE001
                pass
        # the same recomputation for every stage of the step's multi-view pass: algorithmic bytes per launch / measured stage time
        if roofline is not None and batch_stage_ms:
            kern_names = {"preprocess_fwd": "k_preprocess_fwd", "depth_sort": "k_seg_hist/rowscan/scatter x3 (depth bits, range-adaptive digits, per view)",
                          "tile_scan": "k_seg_blocksum + k_seg_totals", "duplicate": "k_seg_duplicate",
                          "tile_sort": "k_seg_hist/rowscan/scatter x2 (tile bits, per view; the last pass also builds the tile ranges)", "tile_ranges": "(fused into the tile sort's last pass)",
                          "render_fwd": "k_render_fwd", "render_bwd": roofline["kernel"],
                          "preprocess_bwd": "k_preprocess_bwd_views<.., FUSE_SH> (one GPU: the SH rows are built in its epilogue; with N > 1 it emits per-view colour gradients and k_sh_grad_combine follows the all-gather)"}
            # bytes the PMC passes saw per step, by stage (profiles/r*_traffic.json: (2 FETCH_SIZE + WRITE_SIZE) KB per launch x launches per
            # step, summed over the stage's kernels). The two sorts share their kernels: their counter bytes exist for "sort_total" only.
            stage_kernels = {"preprocess_fwd": ("k_preprocess_fwd",), "tile_scan": ("k_seg_blocksum", "k_seg_totals", "k_tile_blocksum", "k_tile_scan_blocks"),
                             "duplicate": ("k_seg_duplicate", "k_duplicate"), "tile_ranges": ("k_tile_ranges",), "render_fwd": ("k_render_fwd",),
                             "render_bwd": ("k_render_bwd",), "preprocess_bwd": ("k_preprocess_bwd_views", "k_sh_grad_combine"),
                             "sort_total": ("k_seg_hist", "k_seg_rowscan", "k_seg_scatter", "k_sort_hist", "k_sort_rowscan", "k_sort_scatter")}
            counter_by_stage = {}
            try:
                import glob
                tj_ = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))[-1]))
                if tj_.get("workload") == args.workload and tj_.get("views_per_launch") == G:
                    for stg_, prefixes in stage_kernels.items():
                        tot_ = sum(rec_["hbm_bytes_per_launch_corrected"] * rec_.get("launches_per_step", 0) for kn_, rec_ in tj_["kernels"].items()
                                   if kn_.startswith(prefixes))
                        if tot_ > 0:
                            counter_by_stage[stg_] = tot_
            except Exception:      # noqa: BLE001
                pass
            klist = []
            sort_ms = batch_stage_ms.get("depth_sort", 0.0) + batch_stage_ms.get("tile_sort", 0.0)
            for stg, ms_ in batch_stage_ms.items():
                if stg not in kern_names or ms_ <= 0:
                    continue
                bytes_ = ab.get(stg)
                if stg in ("depth_sort", "tile_sort"):
                    bytes_ = None                              # SURVEY's formula prices the sort as a whole: see "sort_total"
                ent = {"stage": stg, "kernels": kern_names[stg], "ms_per_launch_set": ms_, "views_per_launch": G,
                       "algorithmic_bytes": bytes_ * G if bytes_ else None}
                launch_bytes = bytes_ * G if bytes_ else None
                if stg in ("preprocess_fwd", "preprocess_bwd") and G > 1:
                    # the multi-view pass reads the 236 B/splat of parameters ONCE for its G views: pricing it at G x the per-view formula
                    # would put it above the HBM peak. Bytes the launch has to move: parameters once + the per-(view, splat) arrays.
                    B_sh_ = 12 * (deg + 1) ** 2
                    if stg == "preprocess_fwd":
                        launch_bytes = (44 + B_sh_) * n + 112 * n * G                       # + 64-B record, radii / depth / flags / rect / key / id per (view, splat)
                    else:
                        # rows in (+ re-zeroed: counted once, as SURVEY's formula does); params; geometry + SH gradients out. One GPU (round 6): no
                        # per-view colour gradients leave the kernel; with N > 1 they are written and re-read by the rebuild (12 + 12 B per view and splat)
                        launch_bytes = (48 + (0 if world == 1 else 24)) * n * G + (44 + B_sh_) * n + 44 * n + 12 * n + B_sh_ * n
                    ent["algorithmic_bytes_note"] = "per-launch bytes of the multi-view pass (parameters once); SURVEY's per-view figure x views is in algorithmic_bytes"
                    ent["launch_bytes"] = launch_bytes
                if launch_bytes:
                    ent["achieved_GBps"] = launch_bytes / (ms_ * 1e-3) / 1e9
                    ent["frac_of_hbm_peak"] = ent["achieved_GBps"] / HBM_PEAK_GBPS
                if stg in counter_by_stage:
                    ent["counter_bytes"] = counter_by_stage[stg]
                    ent["frac_by_counters"] = counter_by_stage[stg] / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBPS
                klist.append(ent)
            if sort_ms > 0:
                ent = {"stage": "sort_total", "kernels": "depth sort + tile sort (both segmented by view)", "ms_per_launch_set": sort_ms, "views_per_launch": G,
                       "algorithmic_bytes": ab["sort"] * G, "achieved_GBps": ab["sort"] * G / (sort_ms * 1e-3) / 1e9,
                       "frac_of_hbm_peak": ab["sort"] * G / (sort_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                       "algorithmic_bytes_note": "SURVEY 8(d) prices a sort of 64-bit (tile | depth) keys over all instances; this build sorts the 32 depth "
                                                 "bits over the splats before duplication and only the tile bits over the instances, i.e. it MOVES "
                                                 "far fewer bytes — frac_of_hbm_peak is therefore not a bandwidth figure; frac_by_counters is"}
                if "sort_total" in counter_by_stage:
                    ent["counter_bytes"] = counter_by_stage["sort_total"]
                    ent["frac_by_counters"] = counter_by_stage["sort_total"] / (sort_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS
                klist.append(ent)
            roofline["kernels"] = sorted(klist, key=lambda e_: -e_["ms_per_launch_set"])
            roofline["kernels_note"] = ("per-stage hipEvent spans of the step's own multi-view pass (stage timing on: every call synchronises, kernel "
                                        "durations as in the timed region); algorithmic bytes = SURVEY 8(d) per view x views per launch")
        if strict is not None:
            strict["frac_of_hbm_peak_end_to_end"] = total_bytes / (strict["ms_per_view"] * 1e-3) / 1e9 / HBM_PEAK_GBPS
        stage_table = {}
        for k, v in stage_ms.items():
            stage_table[k] = {"ms": v}
        if "depth_sort" in stage_ms and "tile_sort" in stage_ms:
            stage_table["sort_total"] = {"ms": stage_ms["depth_sort"] + stage_ms["tile_sort"], "algorithmic_bytes": ab["sort"]}
        for k in ab:
            if k in stage_table:
                stage_table[k]["algorithmic_bytes"] = ab[k]
        raster_ms = sum(v for k, v in stage_ms.items()) if stage_ms else None
        rec = {
            "metric": "train views/sec (fwd+bwd raster) at 1M splats 1920x1080" if args.workload == "C3" else f"train views/sec (fwd+bwd raster), workload {args.workload}",
            "value": value, "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None,
            "other_grad_mode": other_mode,
            "strict_single_view": strict, "t_raster_ms_per_step": (ms_per_step - comm_ms["mean"]) if comm_ms else ms_per_step,
            "t_comm_exposed_ms_per_step": comm_ms, "comm_microbench": comm_micro, "clocks": clocks,
            "exchange": exchange_info, "rccl_nranks": exchange_info["rccl_nranks"] if exchange_info else None,
            "per_rank_compute_ms": per_rank_compute_ms, "scaling_model": scaling_model(n, world, VPS, per_rank_compute_ms, ms_per_step),
            "step_ms_p10_p50_p90": step_spread,
            "device": device_info,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {n} splats, {W}x{H}, SH degree {deg}, {GLOBAL_VIEWS} views per iteration "
                                   + ("(weak scaling: fixed per GPU), " if weak else f"sharded over {world} GPU(s) (BASELINE config C4), ") + f"{VPS} view(s) per GPU per step"
                                   + (f" as {K} multi-view pass(es) of {G} view(s) (dvs_raster_forward_views / dvs_raster_backward_*)"
                                      + (" software-pipelined over two contexts/streams, gradients accumulated" if K > 1 else ""))
                                   + ((", RCCL exchange of the gradient rows: " + exchange) if world > 1 else ""),
                       "views_per_step": world * VPS, "views_per_gpu_per_step": VPS, "ms_per_view": ms_per_step / VPS, "absgrad": bool(args.absgrad), "mode": args.mode, "stagger": bool(args.stagger), "groups": K, "views_per_group": G, "early_gather": bool(early), "a9_chunks": len(a9_chunks) if a9_chunks else 1, "async_forward": bool(args.async_forward), "hip_graph": bool(args.graph), "bwd_variant": args.bwd_variant, "fwd_variant": args.fwd_variant, "grad_mode": args.grad_mode, "tile_bounds": "tight (opt-in)" if args.tight_tiles else "canonical", "shN_layout": "tiled[N/64][45][64]" if tiled else "rows[N][45]",
                       "N": n, "V": V, "T": T, "P": Ppix, "tiles": tiles, "sort_passes_p": p},
            "grad_l2_after_exchange": grad_norms,
            "roofline": roofline,
            "pipeline": {"algorithmic_bytes_per_view": total_bytes,
                         "achieved_GBps_end_to_end": total_bytes * VPS / (ms_per_step * 1e-3) / 1e9 if world == 1 else None,
                         "frac_of_hbm_peak_end_to_end": total_bytes * VPS / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS if world == 1 else None,
                         "frac_of_hbm_peak_note": "view-equivalent: SURVEY 8(d)'s per-view bytes x views per step; a multi-view pass reads the "
                                                  "236 B/splat of parameters once for all its views, so the bytes actually moved are fewer — "
                                                  "see counter_bytes_per_step",
                         "counter_bytes_per_step": counter_bytes_step if roofline is not None else None,
                         "frac_of_hbm_peak_by_counters": (counter_bytes_step / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS)
                                                         if (roofline is not None and counter_bytes_step) else None,
                         "sum_of_stage_ms": raster_ms, "stages": stage_table},
        }
        # compute-side figure of SURVEY.md §8(d): pixel-splat interactions I = sum over tiles of (list entries walked before all 256
        # pixels of the tile terminate) x 256, from the saved n_contrib of the last profiled view
        try:
            nc = torch.from_numpy(rast1._d2h(st.n_contrib, (H, W), np.uint32).astype(np.int64))
            ty_, tx_ = st.tiles_y, st.tiles_x
            pad = torch.zeros((ty_ * 16, tx_ * 16), dtype=torch.int64); pad[:H, :W] = nc
            per_tile = pad.view(ty_, 16, tx_, 16).permute(0, 2, 1, 3).reshape(ty_ * tx_, 256).max(dim=1).values
            inter = int(per_tile.sum()) * 256
            rec["interactions"] = {"I_per_view": inter, "mean_list_entries_walked_per_tile": float(per_tile.double().mean()),
                                   "fwd_per_s": inter / (stage_ms["render_fwd"] * 1e-3) if "render_fwd" in stage_ms else None,
                                   "bwd_per_s": inter / (stage_ms["render_bwd"] * 1e-3) if "render_bwd" in stage_ms else None,
                                   "note": "upper bound on evaluated pairs: per-8x8-quadrant culling skips ~2/3 of them (DESIGN.md section 5)"}
            if args.workload == "C3":
                # against the pairs the kernels actually EVALUATE, from the committed launch statistics of this scene:
                # A7 visits a (live entry, 8x8 quadrant) pair with 64 lanes — 72.0 % of the 3.03 M entries per view are live and a live entry
                # reaches 1.87 quadrants (profiles/r05_quadrant_stats.json): 4.08 M visits = 261 M evaluated pairs per view (before early
                # termination of saturated quadrants); A8 walks (entry, 4x4 block) pairs with 16 lanes: 74.9 M pairs per 8-view launch
                # (profiles/r03_a8_probe.txt, TR_STATS) = 149.8 M evaluated pairs per view, 85.4 M of them contributing.
                ev_f, ev_b = 0.720 * 3032201 * 1.87 * 64, 74.9e6 * 16 / 8
                rec["interactions"]["evaluated_pairs_per_view"] = {"fwd": ev_f, "bwd": ev_b, "bwd_contributing": 85.4e6,
                                                                   "source": "profiles/r05_quadrant_stats.json, profiles/r03_a8_probe.txt (launch statistics of this scene)"}
                for key_, ev_ in (("fwd", ev_f), ("bwd", ev_b)):
                    ms_ = batch_stage_ms.get("render_" + key_) if batch_stage_ms else None
                    rec["interactions"]["evaluated_" + key_ + "_per_s"] = (ev_ * G / (ms_ * 1e-3)) if ms_ else None
        except Exception as e:      # noqa: BLE001
            rec["interactions"] = {"error": repr(e)}
        if not args.no_cpu_baseline and world == 1:
            try:
                rec["cpu_baseline"] = cpu_baseline(args.workload)
            except Exception as e:      # the oracle is test infrastructure; its absence must not hide the GPU number
                rec["cpu_baseline"] = {"error": repr(e)}
            # the oracle just computed view 0 at full size: use it as the checker of the HIP path's view 0 (same splats, camera, target)
            ref = cpu_baseline.reference
            if ref is not None:
                try:
                    img_g = rast1.forward(params, cam, sh_degree=deg, absgrad=bool(args.absgrad), out=out, shn_tiled=tiled)
                    g1 = rast1.backward(((img_g - target) * inv_P).contiguous())
                    torch.cuda.synchronize()
                    ok = ~ref["fragile"]
                    ih = img_g.cpu().numpy()
                    err = np.abs(ih[:, ok] - ref["img"][:, ok]) / (1e-4 * np.abs(ref["img"][:, ok]) + 1e-6)
                    par = {"view": 0, "num_rendered_equal": int(rast1.get_num_rendered()) == ref["num_rendered"],
                           "rgb_max_err_over_tol(1e-4 rel + 1e-6)": float(err.max()), "fragile_pixels": int(ref["fragile"].sum())}
                    for k_ in ("pos", "sh0", "shN", "opacity", "scale", "rot"):
                        gk = g1[k_]
                        if k_ == "shN" and tiled:           # back to the reference's [n][45] rows
                            gk = rast1.shn_relayout(gk.contiguous().view(-1), n, to_tiled=False).view(n, 15, 3)
                        a, b = gk.double().cpu().numpy().reshape(-1), np.asarray(ref["grads"][k_], np.float64).reshape(-1)
                        par["grad_rel_l2_" + k_] = float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-300))
                    rec["parity_vs_oracle"] = par
                except Exception as e:      # noqa: BLE001
                    rec["parity_vs_oracle"] = {"error": repr(e)}
        print(json.dumps(rec), flush=True)
    rast1.close()
    for r_ in rasts:
        r_.close()
    barrier()
    if dist is not None:
        dist.destroy_process_group()
    if comm is not None:
        comm.close()
    if watchdog is not None:
        watchdog.cancel()


if __name__ == "__main__":
    main()
