#!/usr/bin/env python3
"""Generate the committed golden fixtures (run in the build container, CPU only):

  python tests/golden/make_golden.py

For each fixture config the scene is regenerated from its seed by the C generator (dvs_synth_*), rendered
by (a) the C++ oracle in fp32 and fp64 and (b) the independent dense PyTorch-autograd formulation
(dense_ref.py, fp64); (a) and (b) must agree (image 1e-9, gradients 1e-7 relative in fp64) before anything
is written. The fixture stores the fp64 oracle's image and gradients (as float32 for size), integer
bin checksums of the fp32 oracle (radii, tiles_touched, sorted keys / values, ranges) and n_contrib.
`make_golden.py edges` regenerates only the edge-case fixture (E_edges.npz: the scenes of edge_scenes.py — everything culled, a splat
on a tile corner, one covering the image, a saturating stack with tile lists > 256, a tile with > 65 536 entries).
"""
import hashlib
import json
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import divshot_amd as dv           # noqa: E402
from oracle import Oracle          # noqa: E402

FIXTURES = {
    # name: (n, W, H, deg, seed, scale_offset, antialias, bg, dense_check, store_arrays)
    "G2_2k_64_deg3": (2000, 64, 64, 3, 1, 0.0, False, (0.3, 0.1, 0.2), True, True),
    "G2b_1500_80x48_deg2_aa": (1500, 80, 48, 2, 4, 0.6, True, (0.0, 0.0, 0.0), True, True),
    "G1_C1_10k_256_deg0": (10000, 256, 256, 0, 1, 0.0, False, (0.0, 0.0, 0.0), False, True),
    "G3_C2_100k_800_deg3": (100000, 800, 800, 3, 1, 0.0, False, (0.0, 0.0, 0.0), False, False),
}
KEYS = ("pos", "sh0", "shN", "opacity", "scale", "rot")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:32]


def build(name):
    n, W, H, deg, seed, soff, aa, bg, dense, store = FIXTURES[name]
    spec = dv.make_spec(n, W, H, sh_degree=deg, seed=seed, scale_log_offset=soff)
    P = dv.synth_splats(spec)
    cam = dv.synth_camera(spec, 0)
    for k in range(3):
        cam.bg[k] = bg[k]
    tgt = dv.synth_target(spec, 0)
    o32, o64 = Oracle(np.float32), Oracle(np.float64)
    img32 = o32.forward(P, cam, sh_degree=deg, antialias=aa)
    img64 = o64.forward(P, cam, sh_degree=deg, antialias=aa)
    dL = ((img64 - tgt) / tgt[0].size)
    g64 = o64.backward(dL)
    meta = {"n": n, "W": W, "H": H, "deg": deg, "seed": seed, "scale_offset": soff, "antialias": aa, "bg": bg,
            "T": int(o32.get("vals").size), "visible": int((o32.get("radii") > 0).sum()),
            "sha_radii": sha(o32.get("radii")), "sha_tiles_touched": sha(o32.get("tiles_touched")),
            "sha_keys": sha(o32.get("keys")), "sha_vals": sha(o32.get("vals")), "sha_ranges": sha(o32.get("ranges")),
            "sha_n_contrib_f32": sha(o32.get("n_contrib")), "fragile_px_f32": int(o32.get("fragile").sum()),
            "img_l2": float(np.linalg.norm(img64)), "grad_l2": {k: float(np.linalg.norm(g64[k])) for k in KEYS}}
    if dense:
        import torch
        import dense_ref
        img_t, leaves = dense_ref.render(P, cam, sh_degree=deg, antialias=aa)
        loss = 0.5 * ((img_t - torch.tensor(tgt, dtype=torch.float64)) ** 2).sum() / tgt[0].size
        loss.backward()
        ok = ~o64.get("fragile").astype(bool)
        d_img = np.abs(img_t.detach().numpy() - img64)[:, ok].max()
        assert d_img < 1e-9, f"{name}: dense torch vs oracle image differ by {d_img}"
        meta["dense_vs_oracle_img_maxabs"] = float(d_img)
        # the dense loss uses its own image, the oracle its own: identical up to 1e-9, so gradients must agree
        for k in KEYS:
            gt = leaves[k].grad.numpy().reshape(g64[k].shape)
            rel = np.abs(gt - g64[k]).max() / (np.abs(g64[k]).max() + 1e-300)
            assert rel < 1e-7, f"{name}: dense torch vs oracle grad {k} differ by {rel} of max"
            meta.setdefault("dense_vs_oracle_grad_rel", {})[k] = float(rel)
    arrays = {}
    if store:
        arrays = {"img": img64.astype(np.float32), "dL": dL.astype(np.float32), "n_contrib": o32.get("n_contrib"),
                  "fragile": np.packbits(o32.get("fragile").astype(bool) | o64.get("fragile").astype(bool))}
        for k in KEYS:
            arrays["g_" + k] = g64[k].astype(np.float32)
    else:
        rng = np.random.default_rng(0)
        idx = rng.choice(n, 64, replace=False)
        arrays = {"sample_idx": idx, "img_sample": img64[:, ::97, ::89].astype(np.float32)}
        for k in KEYS:
            arrays["g_" + k + "_sample"] = g64[k][idx].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
    return meta


def build_edges():
    """The edge cases of SURVEY.md §8(c) (edge_scenes.py: explicit parameter arrays): fp64 oracle outputs, pinned for the scenes the dense
    formulation can hold (E1-E3) by the same cross-check as the seeded goldens. One file, arrays prefixed with the scene name."""
    import edge_scenes as es
    cam = es.camera()
    arrays, metas = {}, {}
    for name, P in es.scenes(cam).items():
        n = P["pos"].shape[0]
        o32, o64 = Oracle(np.float32), Oracle(np.float64)
        img32 = o32.forward(P, cam, sh_degree=es.DEG)
        img64 = o64.forward(P, cam, sh_degree=es.DEG)
        dL = es.upstream(img64.shape)
        g64 = o64.backward(dL.astype(np.float64))
        rng_ = o32.get("ranges")
        meta = {"n": n, "T": int(o32.get("vals").size), "visible": int((o32.get("radii") > 0).sum()),
                "max_tile_list": int((rng_[:, 1].astype(np.int64) - rng_[:, 0]).max()) if rng_.size else 0,
                "min_final_T": float(o32.get("final_T").min()),
                "sha_radii": sha(o32.get("radii")), "sha_tiles_touched": sha(o32.get("tiles_touched")),
                "sha_keys": sha(o32.get("keys")), "sha_vals": sha(o32.get("vals")), "sha_ranges": sha(o32.get("ranges")),
                "fragile_px": int((o32.get("fragile").astype(bool) | o64.get("fragile").astype(bool)).sum()),
                "img_l2": float(np.linalg.norm(img64)), "grad_l2": {k: float(np.linalg.norm(g64[k])) for k in KEYS}}
        if n <= 2000:
            import torch
            import dense_ref
            img_t, leaves = dense_ref.render(P, cam, sh_degree=es.DEG, antialias=False)
            if img_t.requires_grad:              # (an image no splat reaches depends on no parameter)
                (img_t * torch.tensor(dL, dtype=torch.float64)).sum().backward()
            ok = ~o64.get("fragile").astype(bool)
            d_img = np.abs(img_t.detach().numpy() - img64)[:, ok].max() if n else 0.0
            assert d_img < 1e-9, f"{name}: dense torch vs oracle image differ by {d_img}"
            meta["dense_vs_oracle_img_maxabs"] = float(d_img)
            for k in KEYS:
                gt = (leaves[k].grad.numpy() if leaves[k].grad is not None else np.zeros(g64[k].shape)).reshape(g64[k].shape)
                rel = np.abs(gt - g64[k]).max() / (np.abs(g64[k]).max() + 1e-300)
                assert rel < 1e-7, f"{name}: dense torch vs oracle grad {k} differ by {rel} of max"
                meta.setdefault("dense_vs_oracle_grad_rel", {})[k] = float(rel)
        arrays[name + "/img"] = img64.astype(np.float32)
        arrays[name + "/n_contrib"] = o32.get("n_contrib")
        arrays[name + "/fragile"] = np.packbits(o32.get("fragile").astype(bool) | o64.get("fragile").astype(bool))
        if n <= 2000:
            for k in KEYS:
                arrays[name + "/g_" + k] = g64[k].astype(np.float32)
        else:                       # large scene: every 97th row
            for k in KEYS:
                arrays[name + "/g_" + k + "_rows97"] = g64[k][::97].astype(np.float32)
        metas[name] = meta
        print(name, "ok", {k: meta[k] for k in ("T", "visible", "max_tile_list", "min_final_T", "fragile_px")})
    np.savez_compressed(os.path.join(HERE, "E_edges.npz"), **arrays)
    return metas


if __name__ == "__main__":
    only_edges = len(sys.argv) > 1 and sys.argv[1] == "edges"
    mpath = os.path.join(HERE, "manifest.json")
    manifest = json.load(open(mpath)) if only_edges else {}
    if not only_edges:
        for name in FIXTURES:
            manifest[name] = build(name)
            print(name, "ok", {k: v for k, v in manifest[name].items() if k in ("T", "visible", "dense_vs_oracle_img_maxabs")})
    manifest["_edges"] = build_edges()
    with open(mpath, "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
