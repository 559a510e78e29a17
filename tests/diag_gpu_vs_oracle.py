"""Diagnostic dump: per-stage GPU-vs-oracle differences for one config (development aid; lives under tests/ because it
uses the CPU oracle). usage: python tests/diag_gpu_vs_oracle.py N W H DEG [aa]   (env SEED, SOFF)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import divshot_amd as dv
from divshot_amd.raster import Rasterizer, params_to_device
from oracle import Oracle
from util import scene, rel_close, KEYS

n, W, H, deg = [int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (10000, 256, 256, 3))]
aa = len(sys.argv) > 5 and sys.argv[5] == "aa"
seed = int(os.environ.get("SEED", "1")); soff = float(os.environ.get("SOFF", "0"))
spec, P, cam, tgt = scene(n, W, H, deg, seed, scale_offset=soff)
r = Rasterizer(0, max_splats=max(n, 1), max_w=W, max_h=H); r.keep_intermediates(True)
Pd = params_to_device(P, r.tdev)
img = r.forward(Pd, cam, sh_degree=deg, antialias=aa, absgrad=True)
torch.cuda.synchronize()
saved = r.saved(); keys = r.sorted_keys()
o = Oracle(np.float32); ref = o.forward(P, cam, sh_degree=deg, antialias=aa)
print("n", n, "T gpu", r.num_rendered, "T ref", o.get("vals").size, "visible", (saved["radii"] > 0).sum())
for k in ("radii", "tiles_touched", "flags"):
    print(k, "mismatch", (saved[k] != o.get(k)).sum())
for k in ("mean2d", "depth", "conic_opacity", "rgb"):
    a, b = saved[k], o.get(k).reshape(saved[k].shape)
    d = (a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))
    if d.ndim == 1: d = d[:, None]
    print(k, "bit mismatches per column", (d != 0).sum(0), "max ulp", np.abs(d).max(0))
if keys.shape == o.get("keys").shape:
    print("keys mismatch", (keys != o.get("keys")).sum(), "vals mismatch", (saved["vals"] != o.get("vals")).sum(),
          "ranges mismatch", (saved["ranges"] != o.get("ranges")).sum())
else:
    print("keys shape", keys.shape, o.get("keys").shape)
frag = o.get("fragile").astype(bool); ok = ~frag
print("fragile px", frag.sum(), "n_contrib mismatch (non-fragile)", (saved["n_contrib"][ok] != o.get("n_contrib")[ok]).sum())
imgh = img.cpu().numpy()
m, worst = rel_close(imgh[:, ok], ref[:, ok], 1e-4, 1e-6); print("rgb ok frac", m.mean(), "worst", worst, "maxabs", np.abs(imgh - ref).max())
dL = (imgh - tgt) / tgt[0].size
g = r.backward(torch.from_numpy(dL).to(r.tdev), want_mean2d=True); torch.cuda.synchronize()
inter = r.bwd_intermediates()
o64 = Oracle(np.float64); o64.forward(P, cam, sh_degree=deg, antialias=aa)
same = np.array_equal(o64.get("n_contrib"), o.get("n_contrib")) and np.array_equal(o64.get("vals"), o.get("vals"))
print("fp64 oracle same decisions:", same)
if not same: o64 = o
gref = o64.backward(dL)
for k in ("dL_dmean2d", "dL_dconic_opacity", "dL_drgb"):
    m, worst = rel_close(inter[k], o64.get(k), 1e-4, 2e-6); print(k, "ok frac", m.mean(), "worst", worst)
m, worst = rel_close(g["absgrad2d"].cpu().numpy(), o64.get("absgrad"), 1e-4, 2e-6); print("absgrad ok frac", m.mean(), "worst", worst)
for k in KEYS:
    m, worst = rel_close(g[k].cpu().numpy(), gref[k], 1e-4, 2e-6); print("grad", k, "ok frac", m.mean(), "worst", worst)
of32 = o.backward(dL)
for k in KEYS:
    m, worst = rel_close(of32[k], gref[k], 1e-4, 2e-6); print("  (fp32 oracle vs fp64) grad", k, "ok frac", m.mean(), "worst", worst)

# where do the mean2d mismatches live?
got = inter["dL_dmean2d"]; want = o64.get("dL_dmean2d"); w32 = o.get("dL_dmean2d")
m, worst = rel_close(got, want, 1e-4, 1e-5)
bad = np.where(~m.all(1))[0]
print("bad splats", bad.size, "fragile px", frag.sum(), "n_contrib mismatches incl fragile", (saved["n_contrib"] != o.get("n_contrib")).sum())
fy, fx = np.where(frag)
for i in bad[:12]:
    mx, my = saved["mean2d"][i]; rad = saved["radii"][i]
    near = ((np.abs(fx - mx) <= rad) & (np.abs(fy - my) <= rad)).sum() if fx.size else 0
    print(i, "gpu", got[i], "f64", want[i], "f32", w32[i], "rad", rad, "mean", mx, my, "fragile px in footprint", near, "absgrad", o64.get("absgrad")[i])
