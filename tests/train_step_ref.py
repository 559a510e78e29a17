"""CPU restatement of one `train_step` of the gstrain plugin (A10) — TEST INFRASTRUCTURE, not product code.

What the reference's host calls per iteration is one opaque symbol, `train_step(scene)` (application/diverseshot-cli/source/
gs_train.cpp:156; flags main.cpp:24-25,46-48); the trainer behind it is closed source (README.md:46), so this file restates the step
this build's plugin defines (divshot_amd/gstrain/gstrain.cpp trainStep) from independent parts:

  camera draw (xorshift64, one stream) -> oracle forward (oracle/dvs_oracle.hpp) -> photometric gradient of (1 - w) L1 + w (1 - SSIM)
  ((1 - w) sign(out - target) / (3 W H) - w dSSIM/dout; w = --ssim, 0 = L1 only; the SSIM and its gradient restated analytically below)
  -> oracle backward in DVS_GRAD_LINEAGE -> per-group Adam in numpy (beta 0.9 / 0.999, eps 1e-15, bias-corrected) with the learning
  rates of gaussian_trainer_scene.hpp (position: extent x exponential decay) -> abs-grad densification statistics
  (sqrt((|gx| W/2)^2 + (|gy| H/2)^2) per visible splat, denominator + 1).

dtype float32 mirrors the arithmetic the plugin performs; float64 is the ground truth of the same recurrences. The parity test compares
the plugin with the float64 trajectory on every element where the float32 restatement itself stays within tolerance of it: Adam with
eps = 1e-15 turns a gradient that is pure rounding noise into a full-size step of either sign, and such elements are not comparable
between ANY two float32 implementations. The comparable fraction is bounded in the test.
"""
import numpy as np

KEYS = ("pos", "sh0", "shN", "opacity", "scale", "rot")
# gaussian_trainer_scene.hpp (defaults of GaussianTrainConfig; names gs_train.cpp:52-57)
LR = dict(poslrInit=0.00016, poslrFinal=0.0000016, featurelr=0.0025, opacitylr=0.05, scalinglr=0.005, rotationlr=0.001)


def _gauss_window(dt):
    x = np.arange(11, dtype=np.float64) - 5.0
    g = np.exp(-x * x / (2 * 1.5 ** 2))
    return (g / g.sum()).astype(dt)


def _conv_same(img, g):
    """separable 11-tap convolution with zero padding (its own adjoint: the window is symmetric), img [H, W]"""
    H, W = img.shape
    p = np.pad(img, 5)
    t = sum(g[k] * p[:, k:k + W] for k in range(11))
    return sum(g[k] * t[k:k + H, :] for k in range(11))


def ssim_and_grad(x, y):
    """mean SSIM of x against y ([3, H, W]; 11x11 Gaussian window sigma 1.5, zero padding, C1 = 0.01^2, C2 = 0.03^2 — the definition of
    include/dvs_train.h / csrc/ssim.hip, the standard one of the lineage) and its gradient d(mean SSIM)/dx, analytically, in x's dtype.
    With mu1 = g*x, E2 = g*x^2, E12 = g*xy the map is m = A B / (C D), A = 2 mu1 mu2 + C1, B = 2 (E12 - mu1 mu2) + C2,
    C = mu1^2 + mu2^2 + C1, D = (E2 - mu1^2) + sigma2 + C2, and d mean/dx = [ g*(dm/dmu1) + 2 x g*(dm/dE2) + y g*(dm/dE12) ] / N."""
    dt = x.dtype
    g = _gauss_window(dt)
    C1, C2 = dt.type(0.01 ** 2), dt.type(0.03 ** 2)
    two = dt.type(2.0)
    grad = np.zeros_like(x)
    tot = 0.0
    for c in range(3):
        xc, yc = x[c], y[c]
        mu1, mu2 = _conv_same(xc, g), _conv_same(yc, g)
        e2, e12 = _conv_same(xc * xc, g), _conv_same(xc * yc, g)
        s1, s2, s12 = e2 - mu1 * mu1, _conv_same(yc * yc, g) - mu2 * mu2, e12 - mu1 * mu2
        A, B = two * mu1 * mu2 + C1, two * s12 + C2
        Cc, D = mu1 * mu1 + mu2 * mu2 + C1, s1 + s2 + C2
        m = (A * B) / (Cc * D)
        tot += float(m.sum(dtype=np.float64))
        dm_dE12 = two * A / (Cc * D)
        dm_dE2 = -m / D
        # total derivative w.r.t. mu1 at fixed E2, E12: A and C directly, B and D through sigma12 = E12 - mu1 mu2, sigma1 = E2 - mu1^2
        dm_dmu1 = two * mu2 * B / (Cc * D) - two * mu2 * A / (Cc * D) - two * mu1 * m / Cc + two * mu1 * m / D
        grad[c] = _conv_same(dm_dmu1, g) + two * xc * _conv_same(dm_dE2, g) + yc * _conv_same(dm_dE12, g)
    n = x.size
    return tot / n, (grad / dt.type(n)).astype(dt)


def camera_stream(n_cams, count, single_camera=False):
    """The plugin's camera draw: ONE xorshift64 stream shared by all ranks (gstrain.cpp trainStep)."""
    s = 88172645463325252
    out = []
    for _ in range(count):
        s ^= (s << 13) & 0xFFFFFFFFFFFFFFFF
        s ^= s >> 7
        s ^= (s << 17) & 0xFFFFFFFFFFFFFFFF
        out.append(0 if single_camera else s % n_cams)
    return out


def scene_extent(cams):
    """1.1 x the largest distance of a camera centre from their mean; tiny rigs fall back to 5 (gstrain.cpp load_synthetic)."""
    c = np.array([[cam.campos[0], cam.campos[1], cam.campos[2]] for cam in cams], np.float64)
    far = np.sqrt(((c - c.mean(0)) ** 2).sum(1)).max()
    return float(np.float32(1.1 * far)) if far > 1e-3 else 5.0


class TrainStepRef:
    def __init__(self, oracle_cls, cams, targets, init, sh_degree, num_iters, dtype=np.float32, views_per_step=1, world=1, lr=None,
                 ssim_weight=0.0, mcmc_reg=None):
        """ssim_weight: w of L = (1 - w) mean|x - y| + w (1 - mean SSIM) (--ssim, main.cpp:24: 0.2; 0 = L1 only).
        mcmc_reg: (opacity_reg, scale_reg) of densifyStrategy 1 (gstrain.cpp: 0.01, 0.01), added to the summed gradients once per step."""
        self.w_ssim, self.mcmc_reg = float(ssim_weight), mcmc_reg
        self.dt = np.dtype(dtype)
        self.orc = oracle_cls(self.dt)
        self.cams, self.targets = cams, [np.asarray(t, self.dt) for t in targets]
        self.P = {k: np.array(init[k], self.dt) for k in KEYS}
        self.P["shN"] = self.P["shN"].reshape(-1, 15, 3)
        self.M = {k: np.zeros_like(v) for k, v in self.P.items()}
        self.V = {k: np.zeros_like(v) for k, v in self.P.items()}
        self.deg, self.num_iters = sh_degree, num_iters
        self.views, self.world = views_per_step, world
        self.extent = scene_extent(cams)
        self.lr = dict(LR, **(lr or {}))
        self.step = 0
        n = self.P["pos"].shape[0]
        self.order = camera_stream(len(cams), 4096 * views_per_step * world)
        self.grad_accum, self.denom, self.max_radii = np.zeros(n, self.dt), np.zeros(n, self.dt), np.zeros(n, np.int32)
        self.losses = []

    def learning_rates(self):
        f = self.dt.type
        t = min(f(1.0), f(self.step) / f(max(1, self.num_iters)))
        lr_pos = f(self.extent) * np.exp((f(1.0) - t) * np.log(f(self.lr["poslrInit"])) + t * np.log(f(self.lr["poslrFinal"])))
        return {"pos": f(lr_pos), "sh0": f(self.lr["featurelr"]), "shN": f(self.lr["featurelr"]) / f(20.0), "opacity": f(self.lr["opacitylr"]),
                "scale": f(self.lr["scalinglr"]), "rot": f(self.lr["rotationlr"])}

    def gradients(self):
        """Sum over the step's views (all ranks' views: the exchange sums them) of the L1 loss gradient; also the statistics."""
        f = self.dt.type
        n_views = self.views * self.world
        draws = self.order[self.step * n_views:(self.step + 1) * n_views]
        G = {k: np.zeros_like(v) for k, v in self.P.items()}
        loss = 0.0
        for ci in draws:
            cam, tgt = self.cams[ci], self.targets[ci]
            out = self.orc.forward(self.P, cam, sh_degree=self.deg, antialias=False, absgrad=True, grad_mode=1)
            d = out - tgt
            scale = f(1.0) / f(d.size)
            w = f(self.w_ssim)
            dL = (np.sign(d) * (scale * (f(1.0) - w))).astype(self.dt)
            loss += float(np.abs(d).sum() * scale) * (1.0 - self.w_ssim)
            if self.w_ssim > 0:
                val, gs = ssim_and_grad(np.ascontiguousarray(out, self.dt), tgt)
                dL = (dL - w * gs).astype(self.dt)
                loss += self.w_ssim * (1.0 - val)
            g = self.orc.backward(dL, grad_mode=1)
            for k in KEYS:
                G[k] += g[k].reshape(G[k].shape)
            radii = self.orc.get("radii")
            ag = self.orc.get("absgrad")
            vis = radii > 0
            W, H = cam.width, cam.height
            self.grad_accum += np.where(vis, np.hypot(ag[:, 0] * f(0.5 * W), ag[:, 1] * f(0.5 * H)), 0).astype(self.dt)
            self.denom += vis.astype(self.dt)
            self.max_radii = np.maximum(self.max_radii, np.where(vis, radii, 0))
        self.losses.append(loss / len(draws))
        if self.mcmc_reg is not None:            # dvs_mcmc_regularize: d/dlogit of wo mean(sigmoid), d/dlog s of ws mean(exp(log s))
            wo, ws = self.mcmc_reg
            n = self.P["opacity"].shape[0]
            sg = f(1.0) / (f(1.0) + np.exp(-self.P["opacity"]))
            G["opacity"] += (f(wo) / f(n)) * sg * (f(1.0) - sg)
            G["scale"] += (f(ws) / f(3 * n)) * np.exp(self.P["scale"])
        return G

    def adam(self, G):
        f = self.dt.type
        it = self.step + 1
        b1, b2, eps = f(0.9), f(0.999), f(1e-15)
        bc1, bc2 = f(1.0) / (f(1.0) - b1 ** f(it)), f(1.0) / (f(1.0) - b2 ** f(it))
        lr = self.learning_rates()
        for k in KEYS:
            g = G[k]
            if k == "shN":                                   # only the float4 chunks of the active SH bands are stepped (zero gradient above them anyway)
                g = g.copy()
                g[:, (self.deg + 1) ** 2 - 1:, :] = 0
            self.M[k] = b1 * self.M[k] + (f(1.0) - b1) * g
            self.V[k] = b2 * self.V[k] + (f(1.0) - b2) * g * g
            self.P[k] = (self.P[k] - lr[k] * (self.M[k] * bc1) / (np.sqrt(self.V[k] * bc2) + eps)).astype(self.dt)

    def train_step(self):
        G = self.gradients()
        self.adam(G)
        self.step += 1
        return G

    # ---- ADC refinement decision of gstrain.cpp densify() / densify.hip d_action (before any opacity reset) --------------------------
    def adc_actions(self, grow_grad2d, min_opacity=0.005):
        """0 keep, 1 clone, 2 split, 3 prune — and the margin of each decision (relative distance to its nearest threshold)."""
        op = 1.0 / (1.0 + np.exp(-self.P["opacity"].astype(np.float64)))
        smax = np.exp(self.P["scale"].max(1).astype(np.float64))
        avg = np.where(self.denom > 0, self.grad_accum.astype(np.float64) / np.maximum(self.denom, 1), 0.0)
        thr_s = 0.01 * self.extent
        act = np.where(op < min_opacity, 3, np.where(avg >= grow_grad2d, np.where(smax > thr_s, 2, 1), 0))
        margin = np.minimum(np.abs(op - min_opacity) / min_opacity, np.abs(avg - grow_grad2d) / grow_grad2d)
        margin = np.where(avg >= grow_grad2d, np.minimum(margin, np.abs(smax - thr_s) / thr_s), margin)
        return act, margin
