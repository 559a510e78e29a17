"""Probe for the per-quadrant-list form of A8 (GPU box): on one C3 view, how many of a tile's list entries reach the tile
at all (alpha >= 1/255 ellipse vs the 16x16 tile), how many 8x8 quadrants and 4x4 blocks a live entry reaches, and how long a quadrant's
list is against its tile's — the inputs of the cost model in DESIGN.md section 5.5. Exact ellipse-vs-rectangle test in numpy (float64)
on the projected records the HIP forward saved. usage: tools/quadrant_stats.py [n] [W] [H]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import divshot_amd as dv
from divshot_amd.raster import Rasterizer, params_to_device

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
spec = dv.make_spec(n, W, H, sh_degree=3, n_cams=8)
P = dv.synth_splats(spec); cam = dv.synth_camera(spec, 0)
r = Rasterizer(0, max_splats=n, max_w=W, max_h=H)
r.forward(params_to_device(P, r.tdev), cam, sh_degree=3)
torch.cuda.synchronize()
s = r.saved()
rec = s["splat2d"].astype(np.float64)
mx, my, a, b, c, op = rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3], rec[:, 4], rec[:, 5]
tile, sp = s["sorted_tile"].astype(np.int64), s["vals"].astype(np.int64)
tiles_x = (W + 15) // 16
tx, ty = tile % tiles_x, tile // tiles_x
A, B, C, O, MX, MY = a[sp], b[sp], c[sp], op[sp], mx[sp], my[sp]
kappa = 2.0 * np.log(np.maximum(255.0 * O, 1e-300))          # q(d) <= kappa  <=>  opacity exp(-q/2) >= 1/255


def reaches(x0, y0, size):
    """exact: min of a dx^2 + 2 b dx dy + c dy^2 over the pixel-centre rectangle [x0, x0+size-1] x [y0, y0+size-1] <= kappa"""
    X0, X1, Y0, Y1 = x0 - MX, x0 + size - 1 - MX, y0 - MY, y0 + size - 1 - MY
    inx, iny = (X0 <= 0) & (X1 >= 0), (Y0 <= 0) & (Y1 >= 0)
    det = np.maximum(A * C - B * B, 0.0)
    xe = np.where(X0 > 0, X0, X1); ye = np.where(Y0 > 0, Y0, Y1)
    vy = -B / C * xe; d = np.clip(vy, Y0, Y1) - vy; qv = C * d * d + xe * xe * det / C
    hx = -B / A * ye; d2 = np.clip(hx, X0, X1) - hx; qh = A * d2 * d2 + ye * ye * det / A
    q = np.where(inx & iny, 0.0, np.minimum(np.where(inx, np.inf, qv), np.where(iny, np.inf, qh)))
    return q <= kappa


live = reaches(tx * 16.0, ty * 16.0, 16)
quads = np.stack([reaches(tx * 16.0 + 8 * (q & 1), ty * 16.0 + 8 * (q >> 1), 8) for q in range(4)], 1)
blocks = np.stack([reaches(tx * 16.0 + 4 * (k & 3), ty * 16.0 + 4 * (k >> 2), 4) for k in range(16)], 1)
T = tile.size
out = {"n": n, "T": int(T), "live_fraction": float(live.mean()),
       "quadrants_per_live_entry": float(quads[live].sum(1).mean()), "blocks_per_live_entry": float(blocks[live].sum(1).mean()),
       "quadrant_entries_over_tile_entries": float(quads.sum() / T), "quadrant_entries_over_live_entries": float(quads.sum() / live.sum()),
       "blocks_per_quadrant_entry": float(blocks.sum() / quads.sum()),
       "hist_quadrants_per_live_entry": [float((quads[live].sum(1) == k).mean()) for k in range(5)]}
print(json.dumps(out))
