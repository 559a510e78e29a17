"""Independent second implementation of the hot path (oracle validation (ii)):
a DENSE per-pixel formulation in PyTorch (no tiles, no sorting network, gradients by torch.autograd),
run on CPU in the build container only — it is used by make_golden.py to cross-check the C++ oracle and
to produce the committed fixtures; nothing on the GPU box imports it at test time except the CPU-only
golden cross-check. It shares no code with oracle/ or divshot_amd/csrc/.

Conventions restated from the reference's in-tree viewer (fenghuayumo/DIVSHOT):
  gsplat_vs.hlsl:74-110 (EWA, 1.3 tan_fov clamp) · :171-209 (cov3D) · :211-214 (ndc2Pix) · :296-311 (AA, +0.3)
  gsplat_sh.hlsl:42-103 (SH) · gaussian_model.cpp:137-159 (activations) · thresholds SURVEY.md §8(a) A-notes.
"""
import math
import numpy as np
import torch

# the specification's constants are the fp32 roundings of these literals (the HIP path computes in fp32)
def _f32(v):
    return float(np.float32(v))


C0 = _f32(0.28209479177387814)
C1 = _f32(0.4886025119029199)
C2 = [_f32(v) for v in (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)]
C3 = [_f32(v) for v in (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
                        1.445305721320277, -0.5900435899266435)]


def sh_color(deg, sh0, shN, dirs):
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    col = C0 * sh0
    if deg >= 1:
        col = col - C1 * y * shN[:, 0] + C1 * z * shN[:, 1] - C1 * x * shN[:, 2]
    if deg >= 2:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        col = (col + C2[0] * xy * shN[:, 3] + C2[1] * yz * shN[:, 4] + C2[2] * (2 * zz - xx - yy) * shN[:, 5]
               + C2[3] * xz * shN[:, 6] + C2[4] * (xx - yy) * shN[:, 7])
    if deg >= 3:
        col = (col + C3[0] * y * (3 * xx - yy) * shN[:, 8] + C3[1] * xy * z * shN[:, 9]
               + C3[2] * y * (4 * zz - xx - yy) * shN[:, 10] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * shN[:, 11]
               + C3[4] * x * (4 * zz - xx - yy) * shN[:, 12] + C3[5] * z * (xx - yy) * shN[:, 13]
               + C3[6] * x * (xx - 3 * yy) * shN[:, 14])
    return torch.clamp_min(col + 0.5, 0.0)


def render(params, cam, sh_degree=3, antialias=False, dtype=torch.float64, grad_mode=0):
    """params: dict of numpy arrays (A0 layout); cam: ctypes dvs_camera. Returns (image [3,H,W], leaf tensors).
    grad_mode 0 = autograd of the forward as written (DVS_GRAD_TRUE); 1 = DVS_GRAD_LINEAGE (include/dvs_raster.h): the same forward
    values, but the 0.99 alpha cap is a straight-through min and a clamped EWA-Jacobian coordinate is a constant."""
    P = {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True) for k, v in params.items()}
    W, H = cam.width, cam.height
    view = torch.tensor(list(cam.view), dtype=dtype).reshape(4, 4)      # view[c, r]
    proj = torch.tensor(list(cam.proj), dtype=dtype).reshape(4, 4)
    campos = torch.tensor(list(cam.campos), dtype=dtype)
    bg = torch.tensor(list(cam.bg), dtype=dtype)
    pos = P["pos"]
    ones = torch.ones((pos.shape[0], 1), dtype=dtype)
    ph = torch.cat([pos, ones], 1)
    t = ph @ view                                                        # out.r = sum_c in.c * view[c, r]
    hom = ph @ proj
    tz = t[:, 2]
    pw = 1.0 / (hom[:, 3] + float(np.float32(0.0000001)))
    m2x = ((hom[:, 0] * pw + 1.0) * W - 1.0) * 0.5
    m2y = ((hom[:, 1] * pw + 1.0) * H - 1.0) * 0.5
    s = torch.exp(P["scale"])
    q = P["rot"] / P["rot"].norm(dim=1, keepdim=True)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
    M = R * s[:, None, :]
    Sigma = M @ M.transpose(1, 2)
    limx, limy = float(np.float32(1.3)) * cam.tan_fovx, float(np.float32(1.3)) * cam.tan_fovy
    txc = torch.clamp(t[:, 0] / tz, -limx, limx) * tz
    tyc = torch.clamp(t[:, 1] / tz, -limy, limy) * tz
    if grad_mode == 1:      # the lineage multiplies dL/dt.x by 0 on the clamped branch and differentiates t.z at fixed clamped coordinate
        txc = torch.where((t[:, 0] / tz).abs() > limx, txc.detach(), t[:, 0])
        tyc = torch.where((t[:, 1] / tz).abs() > limy, tyc.detach(), t[:, 1])
    zero = torch.zeros_like(tz)
    J = torch.stack([cam.focal_x / tz, zero, -cam.focal_x * txc / (tz * tz),
                     zero, cam.focal_y / tz, -cam.focal_y * tyc / (tz * tz)], 1).reshape(-1, 2, 3)
    Wv = view[:3, :3].T                                                  # Wv[r, c] = d t_r / d p_c
    Tm = J @ Wv
    cov = Tm @ Sigma @ Tm.transpose(1, 2)
    lp = float(np.float32(0.3))
    a, b, c = cov[:, 0, 0] + lp, cov[:, 0, 1], cov[:, 1, 1] + lp
    det = a * c - b * b
    opac = torch.sigmoid(P["opacity"])
    if antialias:
        det_orig = cov[:, 0, 0] * cov[:, 1, 1] - b * b
        opac = opac * torch.sqrt(torch.clamp_min(det_orig / det, 0.0))
    ka, kb, kc = c / det, -b / det, a / det
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp_min(mid * mid - det, float(np.float32(0.1))))
    radius = torch.ceil(3.0 * torch.sqrt(lam)).detach()
    dirs = pos - campos
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    col = sh_color(sh_degree, P["sh0"], P["shN"], dirs)

    tiles_x, tiles_y = (W + 15) // 16, (H + 15) // 16
    with torch.no_grad():
        rminx = torch.clamp(torch.trunc(torch.clamp((m2x - radius) / 16, 0, tiles_x)), 0, tiles_x)
        rmaxx = torch.clamp(torch.trunc(torch.clamp((m2x + radius + 15) / 16, 0, tiles_x)), 0, tiles_x)
        rminy = torch.clamp(torch.trunc(torch.clamp((m2y - radius) / 16, 0, tiles_y)), 0, tiles_y)
        rmaxy = torch.clamp(torch.trunc(torch.clamp((m2y + radius + 15) / 16, 0, tiles_y)), 0, tiles_y)
        visible = (tz > float(np.float32(0.2))) & (det > 0) & (opac > 1.0 / 255.0) & ((rmaxx - rminx) * (rmaxy - rminy) > 0)
        order = torch.argsort(torch.where(visible, tz, torch.full_like(tz, float("inf"))).float(), stable=True)
        order = order[visible[order]]
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dtype), torch.arange(W, dtype=dtype), indexing="ij")
    px, py = xs.reshape(-1, 1), ys.reshape(-1, 1)                        # [P,1]
    o = order
    dx, dy = m2x[o][None, :] - px, m2y[o][None, :] - py                  # [P,V]
    power = -0.5 * (ka[o] * dx * dx + kc[o] * dy * dy) - kb[o] * dx * dy
    with torch.no_grad():
        tile_px, tile_py = torch.floor(px / 16), torch.floor(py / 16)
        in_rect = (tile_px >= rminx[o]) & (tile_px < rmaxx[o]) & (tile_py >= rminy[o]) & (tile_py < rmaxy[o])
    raw = opac[o] * torch.exp(power)
    alpha = torch.clamp_max(raw, float(np.float32(0.99)))
    if grad_mode == 1:      # the lineage's backward uses dL/dG = opacity * dL/dalpha whether or not the cap was hit
        alpha = raw + (alpha - raw).detach()
    keep = in_rect & (power <= 0) & (alpha >= 1.0 / 255.0)
    alpha = torch.where(keep, alpha, torch.zeros_like(alpha))
    T_incl = torch.cumprod(1.0 - alpha, dim=1)
    with torch.no_grad():
        alive = torch.cumprod((T_incl >= float(np.float32(1e-4))).to(dtype), dim=1) > 0     # stops at the first failure
    alpha = torch.where(alive, alpha, torch.zeros_like(alpha))
    T_incl = torch.cumprod(1.0 - alpha, dim=1)
    T_excl = torch.cat([torch.ones_like(T_incl[:, :1]), T_incl[:, :-1]], 1)
    wgt = alpha * T_excl
    img = wgt @ col[o] + T_incl[:, -1:] * bg if o.numel() else bg.expand(W * H, 3)
    return img.T.reshape(3, H, W), P
