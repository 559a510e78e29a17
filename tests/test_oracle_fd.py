"""Oracle validation (i): fp64 central finite differences on every parameter group.

The loss is L = 0.5 * sum((rgb - target)^2) / P, so dL/drgb = (rgb - target)/P — the same upstream
gradient bench.py uses. Finite differences straddling a discontinuity of the forward (alpha < 1/255
skip, T < 1e-4 stop, integer radius / tile rect change, colour clamp) are detected by comparing two
step sizes and excluded; the test asserts that they are rare and that every other element agrees.
"""
import numpy as np
import pytest
import divshot_amd as dv
from oracle import Oracle

KEYS = ("pos", "sh0", "shN", "opacity", "scale", "rot")


def _loss(o, P, cam, tgt, **kw):
    img = o.forward(P, cam, **kw)
    return 0.5 * float(((img - tgt) ** 2).sum()) / tgt[0].size, img


@pytest.mark.parametrize("deg,aa,seed", [(3, False, 1), (3, True, 2), (0, False, 3), (1, True, 4)])
def test_fd_all_groups(deg, aa, seed):
    spec = dv.make_spec(48, 32, 32, sh_degree=deg, seed=seed)
    P32 = dv.synth_splats(spec)
    P = {k: v.astype(np.float64) for k, v in P32.items()}
    # make the splats big enough to overlap several pixels/tiles and each other
    P["scale"] += 0.7
    cam = dv.synth_camera(spec, 0)
    cam.bg[0], cam.bg[1], cam.bg[2] = 0.3, 0.1, 0.2
    tgt = dv.synth_target(spec, 0).astype(np.float64)
    o = Oracle(np.float64)
    kw = dict(sh_degree=deg, antialias=aa)
    L0, img = _loss(o, P, cam, tgt, **kw)
    g = o.backward((img - tgt) / tgt[0].size)
    assert (o.get("radii") > 0).sum() > 30
    total = bad = 0
    rng = np.random.default_rng(seed)
    for k in KEYS:
        flat = P[k].reshape(-1)
        gk = g[k].reshape(-1)
        scale = np.abs(gk).max() + 1e-30
        idxs = np.arange(flat.size) if flat.size <= 400 else rng.choice(flat.size, 400, replace=False)
        for i in idxs:
            old = flat[i]
            fds = []
            for eps in (1e-5, 2.5e-6):
                flat[i] = old + eps; Lp, _ = _loss(o, P, cam, tgt, **kw)
                flat[i] = old - eps; Lm, _ = _loss(o, P, cam, tgt, **kw)
                fds.append((Lp - Lm) / (2 * eps))
            flat[i] = old
            total += 1
            if abs(fds[0] - fds[1]) > 1e-3 * scale + 1e-2 * abs(fds[1]):
                bad += 1          # straddles a discontinuity: the two step sizes disagree
                continue
            assert abs(fds[1] - gk[i]) <= 1e-3 * abs(gk[i]) + 2e-5 * scale, (k, int(i), fds, float(gk[i]))
    assert bad <= 0.02 * total, (bad, total)
