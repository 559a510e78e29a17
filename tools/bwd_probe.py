#!/usr/bin/env python3
"""Time the composite backward (A8) alone on the bench scene under experiment knobs (GPU box):
   python tools/bwd_probe.py [--workload C3] [--reps 20] -- label:VAR=VAL,VAR=VAL ...
Each configuration = a label and environment settings read by the launcher (DVS_BWD_EXTRA_LDS, DVS_MM_DEBUG) plus the pseudo
variables variant=mm|reduce and absgrad=0|1. Prints one JSON line per configuration (mean / min ms over reps)."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("configs", nargs="*")
    a = ap.parse_args()
    cfgs = a.configs or ["tr:variant=tr,fwd=quadrant", "blocks:variant=blocks,fwd=quadrant", "reduce:variant=reduce,fwd=quadrant"]
    if len(cfgs) > 1:          # the launchers read their experiment knobs once per process: one process per configuration
        import subprocess
        for cfg in cfgs:
            env = dict(os.environ)
            for k in ("DVS_BWD_EXTRA_LDS", "DVS_MM_DEBUG", "DVS_TR_DEBUG"):
                env.pop(k, None)
            for x in cfg.partition(":")[2].split(","):
                if "=" in x and x.split("=")[0].isupper():
                    env[x.split("=")[0]] = x.split("=")[1]
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", a.workload, "--reps", str(a.reps), cfg], env=env,
                               capture_output=True, text=True)
            sys.stdout.write("".join(l + "\n" for l in p.stdout.splitlines() if l.startswith("{")) or p.stderr[-2000:])
            sys.stdout.flush()
        return
    import numpy as np, torch
    import divshot_amd as dv
    from divshot_amd.raster import Rasterizer, params_to_device
    from bench import WORKLOADS
    n, W, H, deg, soff = WORKLOADS[a.workload]
    spec = dv.make_spec(n, W, H, sh_degree=deg, n_cams=8, scale_log_offset=soff)
    dev = torch.device("cuda", 0)
    P = params_to_device(dv.synth_splats(spec), dev)
    cam = dv.synth_camera(spec, 0)
    tgt = torch.from_numpy(dv.synth_target(spec, 0)).to(dev)
    r = Rasterizer(0, max_splats=n, max_w=W, max_h=H)
    for cfg in cfgs:
        label, _, rest = cfg.partition(":")
        kv = dict(x.split("=") for x in rest.split(",") if x)
        variant, absgrad, fvar = kv.pop("variant", "blocks"), int(kv.pop("absgrad", "1")), kv.pop("fwd", "blocks")
        for k in ("DVS_BWD_EXTRA_LDS", "DVS_MM_DEBUG", "DVS_TR_DEBUG"):
            os.environ.pop(k, None)
        os.environ.update(kv)
        r.set_backward_variant(variant)
        r.set_forward_variant(fvar)
        fms = []
        for i in range(a.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            r.enable_timing(True)
            img = r.forward(P, cam, sh_degree=deg, absgrad=bool(absgrad))
            fms.append(r.stage_timing().get("render_fwd", 0.0))
            r.enable_timing(False)
        img = r.forward(P, cam, sh_degree=deg, absgrad=bool(absgrad))
        dL = ((img - tgt) / (W * H)).contiguous()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.reps + 1)]
        for _ in range(3):
            r.backward_composite(dL); r.backward_project()
        torch.cuda.synchronize()
        ms = []
        for i in range(a.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); r.backward_composite(dL); e1.record()
            r.backward_project()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        print(json.dumps({"label": label, "variant": variant, "absgrad": absgrad, "env": kv, "mean_ms": float(np.mean(ms)), "min_ms": float(np.min(ms)), "fwd_variant": fvar, "render_fwd_ms": float(np.mean(fms[2:]))}), flush=True)


if __name__ == "__main__":
    main()
