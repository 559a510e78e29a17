"""CPU restatement of one `train_step` of the gstrain plugin (A10) — TEST INFRASTRUCTURE, not product code.

What the reference's host calls per iteration is one opaque symbol, `train_step(scene)` (application/diverseshot-cli/source/
gs_train.cpp:156; flags main.cpp:24-25,46-48); the trainer behind it is closed source (README.md:46), so this file restates the step
this build's plugin defines (divshot_amd/gstrain/gstrain.cpp trainStep) from independent parts:

  camera draw (xorshift64, one stream) -> oracle forward (oracle/dvs_oracle.hpp) -> L1 photometric gradient sign(out - target) / (3 W H)
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
LR = dict(poslrInit=0.00016, poslrFinal=0.0000016, featurelr=0.0025, opacitylr=0.05, scalinglr=0.001, rotationlr=0.001)


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
    def __init__(self, oracle_cls, cams, targets, init, sh_degree, num_iters, dtype=np.float32, views_per_step=1, world=1, lr=None):
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
            dL = (np.sign(d) * scale).astype(self.dt)
            loss += float(np.abs(d).sum() * scale)
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
