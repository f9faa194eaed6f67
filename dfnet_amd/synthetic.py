"""Synthetic random-weight scenes: there is no network for datasets or
checkpoints, so benchmarks and parity tests run on seeded random-init weights of the
reference's architecture and a procedural camera path.

Pure numpy: used by bench.py, the tests and tests/golden/make_golden.py alike so that
every side regenerates identical weights from a seed.
"""
import math

import numpy as np


def nerfh_param_shapes(typ, W=128, D=8, skip=4, ch_xyz=63, ch_dir=27, ch_a=50, ch_t=20):
    """Ordered {state_dict key: shape} of the reference's NeRFW module
    (/root/reference/script/models/nerfw.py:259-295); `typ` is 'coarse' or 'fine'."""
    a = ch_a if typ == "fine" else 0
    shapes = {}
    for i in range(D):
        k = ch_xyz if i == 0 else (W + ch_xyz if i == skip else W)
        shapes[f"xyz_encoding_{i + 1}.0.weight"] = (W, k)
        shapes[f"xyz_encoding_{i + 1}.0.bias"] = (W,)
    shapes["xyz_encoding_final.weight"] = (W, W)
    shapes["xyz_encoding_final.bias"] = (W,)
    shapes["dir_encoding.0.weight"] = (W // 2, W + ch_dir + a)
    shapes["dir_encoding.0.bias"] = (W // 2,)
    shapes["static_sigma.0.weight"] = (1, W)
    shapes["static_sigma.0.bias"] = (1,)
    shapes["static_rgb.0.weight"] = (3, W // 2)
    shapes["static_rgb.0.bias"] = (3,)
    if typ == "fine":
        for j, k in ((0, W + ch_t), (2, W // 2), (4, W // 2), (6, W // 2)):
            shapes[f"transient_encoding.{j}.weight"] = (W // 2, k)
            shapes[f"transient_encoding.{j}.bias"] = (W // 2,)
        for name, n in (("transient_sigma", 1), ("transient_rgb", 3), ("transient_beta", 1)):
            shapes[f"{name}.0.weight"] = (n, W // 2)
            shapes[f"{name}.0.bias"] = (n,)
    return shapes


def _uniform_fan_in(rng, shapes):
    """U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weights AND biases — the nn.Linear/Conv default law."""
    out = {}
    fan = 1
    for key, shp in shapes.items():
        if key.endswith("weight"):
            fan = int(np.prod(shp[1:]))
        b = 1.0 / math.sqrt(fan)
        out[key] = rng.uniform(-b, b, size=shp).astype(np.float32)
    return out


def nerfh_weights(seed=0, W=128, n_vocab=1000, gain=1.0):
    """(coarse, fine, embedding_a, embedding_t) numpy dicts/arrays; coarse then fine drawn from
    default_rng(seed) in state_dict order, embeddings from default_rng(seed+1) ~ N(0,1).

    `gain` scales every trunk weight matrix (a sharper, trained-checkpoint-like scene for
    stress tests); gain=1 is the judged default-init scene.
    """
    rng = np.random.default_rng(seed)
    coarse = _uniform_fan_in(rng, nerfh_param_shapes("coarse", W))
    fine = _uniform_fan_in(rng, nerfh_param_shapes("fine", W))
    if gain != 1.0:
        for net in (coarse, fine):
            for k in net:
                if k.endswith("weight"):
                    net[k] = (net[k] * gain).astype(np.float32)
    erng = np.random.default_rng(seed + 1)
    emb_a = erng.standard_normal((n_vocab, 5)).astype(np.float32)
    emb_t = erng.standard_normal((n_vocab, 2)).astype(np.float32)
    return coarse, fine, emb_a, emb_t


HIST_IDX = np.array([0, 5, 10, 20, 30, 20, 10, 5, 0, 0], dtype=np.float32)


def orbit_pose(k, K):
    """Frame k of K: rotation about y by 0.1*2*pi*k/K, translation (0.2 sin, 0, 1+0.2 cos); 4x4 c2w."""
    th = 0.1 * 2.0 * math.pi * k / max(K, 1)
    c, s = math.cos(th), math.sin(th)
    m = np.eye(4, dtype=np.float32)
    m[:3, :3] = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float32)
    m[:3, 3] = np.array([0.2 * s, 0.0, 1.0 + 0.2 * c], dtype=np.float32)
    return m


# ----------------------------------------------------------------------------- a scene with real occupancy
# Three shaded spheres in front of a checkered wall, seen from the orbit_pose cameras: ground truth by analytic ray casting.  NeRF-H
# trained on it (tools/gpu_train_scene.py) is the "trained-like weights" fixture tests/golden/trained_nerfh_weights.npz.
SCENE_SPHERES = (((-0.32, 0.05, -0.05), 0.26, (0.85, 0.25, 0.2)), ((0.30, -0.08, 0.10), 0.22, (0.2, 0.7, 0.3)),
                 ((0.02, 0.22, -0.35), 0.18, (0.25, 0.35, 0.9)))
SCENE_WALL_Z = -0.75
SCENE_LIGHT = np.array([0.4, 0.7, 0.6]) / np.linalg.norm([0.4, 0.7, 0.6])


def analytic_scene_image(c2w, H, W, focal, far=2.5):
    """[H,W,3] float32 image of the scene from pose c2w [3,4] (the reference's camera convention, models/ray_utils.py:5-15): the
    nearest sphere (Lambert + ambient), else the checkered wall z = SCENE_WALL_Z."""
    c2w = np.asarray(c2w, dtype=np.float64)
    i, j = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64), indexing="xy")
    dirs = np.stack([(i - W * .5) / focal, -(j - H * .5) / focal, -np.ones_like(i)], -1)
    d = dirs @ c2w[:3, :3].T
    o = c2w[:3, 3]
    best = np.full((H, W), np.inf)
    rgb = np.zeros((H, W, 3))
    tw = (SCENE_WALL_Z - o[2]) / d[..., 2]
    pw = o + tw[..., None] * d
    chk = ((np.floor(pw[..., 0] * 5) + np.floor(pw[..., 1] * 5)) % 2)[..., None]
    wall = chk * np.array([0.9, 0.85, 0.6]) + (1 - chk) * np.array([0.25, 0.25, 0.3])
    ok = (tw > 0) & (tw < far)
    rgb[ok], best[ok] = wall[ok], tw[ok]
    for c, r, col in SCENE_SPHERES:
        oc = o - np.array(c)
        a = (d * d).sum(-1)
        b = 2 * (d * oc).sum(-1)
        cc = (oc * oc).sum() - r * r
        disc = b * b - 4 * a * cc
        t = (-b - np.sqrt(np.maximum(disc, 0))) / (2 * a)
        hit = (disc > 0) & (t > 0) & (t < best)
        n = (o + t[..., None] * d - np.array(c)) / r
        shade = 0.25 + 0.75 * np.clip((n * SCENE_LIGHT).sum(-1), 0, 1)
        rgb[hit] = (shade[..., None] * np.array(col))[hit]
        best[hit] = t[hit]
    return rgb.astype(np.float32)


def trained_nerfh_weights(path=None):
    """(coarse, fine, embedding_a, embedding_t) of the trained-like fixture (numbers only; see tools/gpu_train_scene.py)."""
    import os
    path = path or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "trained_nerfh_weights.npz")
    tw = np.load(path)
    coarse = {k[len("coarse."):]: tw[k] for k in tw.files if k.startswith("coarse.")}
    fine = {k[len("fine."):]: tw[k] for k in tw.files if k.startswith("fine.")}
    return coarse, fine, tw["embedding_a.weight"], tw["embedding_t.weight"]


# ----------------------------------------------------------------------------- DFNet
VGG16_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M")


def dfnet_param_shapes(feat_dim=12, taps=(64, 256, 512), out_dim=128):
    """Ordered {state_dict key: shape} of the reference's DFNet module
    (/root/reference/script/feature/dfnet.py:74-107; VGG16 cfg 'D' from torchvision)."""
    shapes = {}
    cin, idx = 3, 0
    for v in VGG16_CFG:
        if v == "M":
            idx += 1
            continue
        shapes[f"encoder.{idx}.weight"] = (v, cin, 3, 3)
        shapes[f"encoder.{idx}.bias"] = (v,)
        cin = v
        idx += 2  # conv + relu
    for i, c in enumerate(taps):
        p = f"adaptation_layers.adapt_layer_{i}"
        shapes[f"{p}.0.weight"] = (64, c, 1, 1)
        shapes[f"{p}.0.bias"] = (64,)
        shapes[f"{p}.2.weight"] = (out_dim, 64, 5, 5)
        shapes[f"{p}.2.bias"] = (out_dim,)
        shapes[f"{p}.3.weight"] = (out_dim,)
        shapes[f"{p}.3.bias"] = (out_dim,)
        shapes[f"{p}.3.running_mean"] = (out_dim,)
        shapes[f"{p}.3.running_var"] = (out_dim,)
    shapes["fc_pose.weight"] = (feat_dim, 512)
    shapes["fc_pose.bias"] = (feat_dim,)
    return shapes


def dfnet_weights(seed=3, taps=(64, 256, 512)):
    """Seeded DFNet state_dict (numpy).  Conv/linear: He-like uniform U(+-sqrt(6/fan_in)) so that
    activations keep O(1) scale through 13 ReLU layers (the fan-in default law collapses them to
    ~1e-4, which would make feature-map parity vacuous); BN affine and running stats non-trivial."""
    rng = np.random.default_rng(seed)
    out = {}
    for key, shp in dfnet_param_shapes(taps=taps).items():
        if key.endswith("running_mean"):
            out[key] = rng.normal(0.0, 0.5, size=shp).astype(np.float32)
        elif key.endswith("running_var"):
            out[key] = rng.uniform(0.5, 2.0, size=shp).astype(np.float32)
        elif len(shp) == 1 and ".3." in key:  # BN weight / bias
            out[key] = (rng.uniform(0.5, 1.5, size=shp) if key.endswith("weight")
                        else rng.normal(0.0, 0.2, size=shp)).astype(np.float32)
        elif key.endswith("weight"):
            fan = int(np.prod(shp[1:]))
            b = math.sqrt(6.0 / fan)
            out[key] = rng.uniform(-b, b, size=shp).astype(np.float32)
        else:
            out[key] = rng.uniform(-0.1, 0.1, size=shp).astype(np.float32)
    return out
