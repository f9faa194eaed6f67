"""CPU oracle for the NeRF-H render hot path.  TEST INFRASTRUCTURE ONLY.

This module is a from-scratch CPU (torch fp32) restatement of the reference's
NeRF-H volumetric rendering path.  It exists to CHECK the HIP kernels and to
provide the `cpu_baseline` leg of bench.py.  Only `tests/`,
`__graft_entry__.smoke()` and bench.py's cpu_baseline may import it; nothing
under `dfnet_amd/` does (the product path fails loudly without its HIP
library instead of falling back to this file).

Parity pinning: every function here is checked against outputs captured from
the reference's own modules (imported from /root/reference in the build
container by tests/golden/make_golden.py) — see tests/test_oracle_golden.py.

All `file:line` citations are relative to /root/reference/script/.

Conventions: parameters are plain dicts {state_dict key: tensor}, with the
reference's key names (`xyz_encoding_1.0.weight`, `static_sigma.0.bias`, ...,
models/nerfw.py:259-295).
"""
import math

import torch
import torch.nn.functional as F

N_RAW = 9  # fine net channels: rgb_s(3) sigma_s rgb_t(3) sigma_t beta_t (models/nerfw.py:340-354)


# --------------------------------------------------------------------------- rays
def get_rays(H, W, focal, c2w):
    """Pinhole rays of an H x W image (models/ray_utils.py:5-15).

    Pixel centres sit at integer coordinates (no +0.5); rays_d is NOT
    normalised; rays_o is the camera centre for every pixel.  Returns
    (rays_o, rays_d), both [H, W, 3].
    """
    c2w = torch.as_tensor(c2w, dtype=torch.float32)
    col = torch.linspace(0, W - 1, W)  # pixel x
    row = torch.linspace(0, H - 1, H)  # pixel y
    cx = ((col - W * .5) / focal)[None, :].expand(H, W)
    cy = (-(row - H * .5) / focal)[:, None].expand(H, W)
    cam = torch.stack([cx, cy, -torch.ones(H, W)], -1)  # camera-frame direction
    # rays_d[h,w,a] = sum_b cam[h,w,b] * c2w[a,b]   (row-vector * R^T, ray_utils.py:12)
    rays_d = torch.sum(cam[..., None, :] * c2w[:3, :3], -1)
    rays_o = c2w[:3, 3].expand(rays_d.shape)
    return rays_o, rays_d


def pack_ray_rows(rays_o, rays_d, near, far, img_idx):
    """The 21-float ray row [o d near far viewdir hist(10)] (models/rendering.py:364-389)."""
    d = rays_d.reshape(-1, 3).float()
    o = rays_o.reshape(-1, 3).float()
    view = d / torch.norm(d, dim=-1, keepdim=True)
    nf = torch.ones_like(d[:, :1])
    idx = torch.as_tensor(img_idx, dtype=torch.float32)
    if idx.dim() == 1:
        idx = idx[None]
    if idx.shape[0] != d.shape[0]:
        idx = idx.repeat(d.shape[0], 1)
    return torch.cat([o, d, near * nf, far * nf, view, idx], 1)


# --------------------------------------------------------------------------- encoding
def posenc(x, L):
    """[x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)] (models/nerfw.py:105-133).

    Each block is as wide as x; the bands are exact powers of two
    (2**linspace(0, L-1, L), nerfw.py:117).
    """
    parts = [x]
    for k in range(L):
        f = float(2 ** k)
        parts.append(torch.sin(x * f))
        parts.append(torch.cos(x * f))
    return torch.cat(parts, -1)


def hist_embed(table, hist_idx):
    """nn.Embedding lookup of the 10 histogram bins, flattened bin-major (models/nerfw.py:69-78)."""
    rows = table[hist_idx.long()]  # [N, bins, dim]
    return rows.reshape(rows.shape[0], -1)


# --------------------------------------------------------------------------- network
def _lin(p, name, x):
    return F.linear(x, p[name + ".weight"], p[name + ".bias"])


def nerfh_trunk(p, pe_xyz, D=8, skip=4):
    """The D-layer ReLU trunk with [input, h] concat before layer index `skip` (models/nerfw.py:326-330)."""
    h = pe_xyz
    for i in range(D):
        if i == skip:
            h = torch.cat([pe_xyz, h], -1)
        h = torch.relu(_lin(p, f"xyz_encoding_{i + 1}.0", h))
    return h


def nerfh_sigma(p, pe_xyz, D=8):
    """Coarse test-time query: sigma only (models/nerfw.py:315-334); Softplus is inside the net."""
    return F.softplus(_lin(p, "static_sigma.0", nerfh_trunk(p, pe_xyz, D)))


def nerfh_static(p, pe_xyz, pe_dir_a, D=8):
    """[rgb_s(3), sigma_s]: the `output_transient=False` branch (models/nerfw.py:336-343)."""
    h = nerfh_trunk(p, pe_xyz, D)
    sigma = F.softplus(_lin(p, "static_sigma.0", h))
    fin = _lin(p, "xyz_encoding_final", h)
    dire = torch.relu(_lin(p, "dir_encoding.0", torch.cat([fin, pe_dir_a], -1)))
    rgb = torch.sigmoid(_lin(p, "static_rgb.0", dire))
    return torch.cat([rgb, sigma], -1)


def nerfh_fine(p, pe_xyz, pe_dir, a, t, D=8):
    """Fine NeRF-H query -> [.., 9] (models/nerfw.py:297-354)."""
    h = nerfh_trunk(p, pe_xyz, D)
    sigma = F.softplus(_lin(p, "static_sigma.0", h))
    fin = _lin(p, "xyz_encoding_final", h)
    dire = torch.relu(_lin(p, "dir_encoding.0", torch.cat([fin, pe_dir, a], -1)))
    rgb = torch.sigmoid(_lin(p, "static_rgb.0", dire))
    tr = torch.cat([fin, t], -1)
    for j in (0, 2, 4, 6):
        tr = torch.relu(_lin(p, f"transient_encoding.{j}", tr))
    t_sigma = F.softplus(_lin(p, "transient_sigma.0", tr))
    t_rgb = torch.sigmoid(_lin(p, "transient_rgb.0", tr))
    t_beta = F.softplus(_lin(p, "transient_beta.0", tr))
    return torch.cat([rgb, sigma, t_rgb, t_sigma, t_beta], -1)


def query_coarse_sigma(p, pts, L_xyz=10, netchunk=65536):
    """run_network_NeRFW, typ='coarse', test_time (models/nerfw.py:37-46): [R,N,3] -> [R,N,1]."""
    flat = pts.reshape(-1, 3)
    out = [nerfh_sigma(p, posenc(flat[i:i + netchunk], L_xyz)) for i in range(0, flat.shape[0], netchunk)]
    return torch.cat(out, 0).reshape(*pts.shape[:-1], 1)


def query_fine(p, emb_a, emb_t, pts, viewdirs, hist_idx, L_xyz=10, L_dir=4, netchunk=65536):
    """run_network_NeRFW, typ='fine' (models/nerfw.py:62-95): [R,N,3] -> [R,N,9]."""
    R, N = pts.shape[:2]
    flat = pts.reshape(-1, 3)
    dirs = viewdirs[:, None].expand(pts.shape).reshape(-1, 3)
    a = hist_embed(emb_a, hist_idx).repeat_interleave(N, 0)  # 'n1 c -> (n1 n2) c'
    t = hist_embed(emb_t, hist_idx).repeat_interleave(N, 0)
    out = []
    for i in range(0, flat.shape[0], netchunk):
        s = slice(i, i + netchunk)
        out.append(nerfh_fine(p, posenc(flat[s], L_xyz), posenc(dirs[s], L_dir), a[s], t[s]))
    return torch.cat(out, 0).reshape(R, N, N_RAW)


# --------------------------------------------------------------------------- sampling / compositing
def _deltas(z):
    """z[i+1]-z[i] with a last interval of 1e2 and no |d| factor (models/rendering.py:161-166)."""
    return torch.cat([z[:, 1:] - z[:, :-1], torch.full_like(z[:, :1], 1e2)], -1)


def _excl_cumprod(one_minus_alpha):
    """T_i = prod_{j<i} (1-alpha_j), no epsilon inside (models/rendering.py:176-178)."""
    shifted = torch.cat([torch.ones_like(one_minus_alpha[:, :1]), one_minus_alpha], -1)
    return torch.cumprod(shifted[:, :-1], -1)


def coarse_weights(sigma, z, noise=None):
    """Coarse test-time compositing: alpha = 1-exp(-delta*relu(sigma+noise)) (models/rendering.py:173-193).

    Returns (acc, weights).  `noise` stands for randn*raw_noise_std (zero at test time).
    """
    s = sigma if noise is None else sigma + noise
    alpha = 1 - torch.exp(-_deltas(z) * torch.relu(s))
    w = alpha * _excl_cumprod(1 - alpha)
    return w.sum(-1), w


def sample_pdf(bins, weights, n, det=True, u=None):
    """Inverse-CDF importance sampling (models/rendering.py:24-65).

    `u` overrides the uniform draws (shape [R, n]) for the non-deterministic mode.
    """
    w = weights + 1e-5
    pdf = w / w.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, -1)], -1)
    if u is None:
        if det:
            u = torch.linspace(0., 1., n).expand(cdf.shape[0], n)
        else:
            u = torch.rand(cdf.shape[0], n)
    u = u.contiguous()
    hi = torch.searchsorted(cdf, u, right=True)
    lo = (hi - 1).clamp(min=0)
    hi = hi.clamp(max=cdf.shape[-1] - 1)
    c_lo, c_hi = cdf.gather(1, lo), cdf.gather(1, hi)
    b_lo, b_hi = bins.gather(1, lo), bins.gather(1, hi)
    den = c_hi - c_lo
    den = torch.where(den < 1e-5, torch.ones_like(den), den)
    return b_lo + (u - c_lo) / den * (b_hi - b_lo)


def composite_fine(raw, z, beta_min=0.1, white_bkgd=False, test_time=True, static_only=True):
    """Static + transient compositing of the 9-channel fine output (models/rendering.py:144-243).

    Returns dict(rgb, disp, acc, weights, depth, transient_sigmas, beta).  At
    test_time & static_only the reference returns the JOINT rgb (static+transient
    under the joint transmittance), a depth from the STATIC-ONLY transmittance and
    disp = 1/max(1e-10, depth/sum(joint w)) (rendering.py:211-230, quirk Q2).
    """
    rgb_s, sig_s = raw[..., 0:3], raw[..., 3]
    rgb_t, sig_t, beta_t = raw[..., 4:7], raw[..., 7], raw[..., 8]
    d = _deltas(z)
    a_s = 1 - torch.exp(-d * sig_s)
    a_t = 1 - torch.exp(-d * sig_t)
    a = 1 - torch.exp(-d * (sig_s + sig_t))
    T = _excl_cumprod(1 - a)
    w_s, w_t, w = a_s * T, a_t * T, a * T
    acc = w.sum(-1)
    map_s = (w_s[..., None] * rgb_s).sum(-2)
    if white_bkgd:
        map_s = map_s + (1 - acc[:, None])
    map_t = (w_t[..., None] * rgb_t).sum(-2)
    beta = (w_t * beta_t).sum(-1) + beta_min
    rgb = map_s + map_t
    if test_time and static_only:
        w_so = a_s * _excl_cumprod(1 - a_s)
        depth = (w_so * z).sum(-1)
    else:
        depth = (w * z).sum(-1)
    disp = 1. / torch.max(1e-10 * torch.ones_like(depth), depth / w.sum(-1))
    return dict(rgb=rgb, disp=disp, acc=acc, weights=w, depth=depth, transient_sigmas=sig_t, beta=beta)


# --------------------------------------------------------------------------- render
def coarse_z(near, far, n, R, lindisp=False):
    """z = near(1-t) + far t, t = linspace(0,1,n); lindisp: linear in disparity, z = 1/((1-t)/near + t/far)
    (models/rendering.py:269-275)."""
    t = torch.linspace(0., 1., n)
    if lindisp:
        return (1. / (1. / near * (1. - t) + 1. / far * t)).expand(R, n)
    return (near * (1. - t) + far * t).expand(R, n)


def render_rays(rows, coarse, fine, emb_a, emb_t, Nc, Ni, netchunk=65536, retraw=False, stages=None, lindisp=False):
    """Test-time render of packed 21-float ray rows (models/rendering.py:245-337 with
    perturb=0, raw_noise_std=0, white_bkgd=False, test_time=True).  white_bkgd=True is not a working option of this
    path in the reference: rendering.py:295 hands it to the coarse compositor as `output_transient`, which raises TypeError
    (tests/golden/make_golden.py, G14, records that).

    `stages`, if a dict, receives the intermediate tensors for stage-level parity tests.
    """
    o, d = rows[:, 0:3], rows[:, 3:6]
    near, far = rows[:, 6:7], rows[:, 7:8]
    view, hist = rows[:, 8:11], rows[:, 11:]
    R = rows.shape[0]
    z = coarse_z(near, far, Nc, R, lindisp)
    pts = o[:, None, :] + d[:, None, :] * z[..., None]
    sig = query_coarse_sigma(coarse, pts, netchunk=netchunk)[..., 0]
    _, w = coarse_weights(sig, z)
    mid = .5 * (z[:, 1:] + z[:, :-1])
    zs = sample_pdf(mid, w[:, 1:-1], Ni, det=True).detach()  # rendering.py:302: no gradient through the sampler
    zf, _ = torch.sort(torch.cat([z, zs], -1), -1)
    pts_f = o[:, None, :] + d[:, None, :] * zf[..., None]
    raw = query_fine(fine, emb_a, emb_t, pts_f, view, hist, netchunk=netchunk)
    out = composite_fine(raw, zf)
    if stages is not None:
        stages.update(z_coarse=z, sigma_coarse=sig, weights_coarse=w, z_samples=zs, z_fine=zf, raw=raw)
    ret = dict(rgb_map=out["rgb"], disp_map=out["disp"], acc_map=out["acc"])
    if retraw:
        ret["raw"] = raw
    return ret


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """Normalised-device-coordinate rays of a forward-facing scene (models/ray_utils.py:27-46): origins moved to the near
    plane, then the projective map of the NeRF paper's appendix (o' and d' such that o' + t' d' covers [-1, 1]^3)."""
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    o = rays_o + t[..., None] * rays_d
    sx, sy = -1. / (W / (2. * focal)), -1. / (H / (2. * focal))
    o_ndc = torch.stack([sx * o[..., 0] / o[..., 2], sy * o[..., 1] / o[..., 2], 1. + 2. * near / o[..., 2]], -1)
    d_ndc = torch.stack([sx * (rays_d[..., 0] / rays_d[..., 2] - o[..., 0] / o[..., 2]),
                         sy * (rays_d[..., 1] / rays_d[..., 2] - o[..., 1] / o[..., 2]), -2. * near / o[..., 2]], -1)
    return o_ndc, d_ndc


def render(H, W, focal, chunk, coarse, fine, emb_a, emb_t, Nc, Ni, near, far, img_idx,
           c2w=None, rays=None, netchunk=65536, ndc=False, lindisp=False, c2w_staticcam=None):
    """render() at test time (models/rendering.py:353-400): returns [rgb, disp, acc] shaped like the rays.  View directions
    come from the rays BEFORE the static-camera substitution and the NDC map (rendering.py:364-376)."""
    if c2w is not None:
        rays_o, rays_d = get_rays(H, W, focal, torch.as_tensor(c2w, dtype=torch.float32)[:3, :4])
    else:
        rays_o, rays_d = rays
    view = rays_d.reshape(-1, 3).float()
    view = view / torch.norm(view, dim=-1, keepdim=True)
    if c2w_staticcam is not None:
        rays_o, rays_d = get_rays(H, W, focal, torch.as_tensor(c2w_staticcam, dtype=torch.float32)[:3, :4])
    sh = rays_d.shape
    if ndc:
        rays_o, rays_d = ndc_rays(H, W, focal, 1., rays_o, rays_d)
    rows = pack_ray_rows(rays_o, rays_d, near, far, img_idx)
    rows[:, 8:11] = view
    outs = [render_rays(rows[i:i + chunk], coarse, fine, emb_a, emb_t, Nc, Ni, netchunk, lindisp=lindisp)
            for i in range(0, rows.shape[0], chunk)]
    cat = {k: torch.cat([o[k] for o in outs], 0) for k in outs[0]}
    return [cat[k].reshape(*sh[:-1], *cat[k].shape[1:]) for k in ("rgb_map", "disp_map", "acc_map")]


# --------------------------------------------------------------------------- training mode
def query_coarse_static(p, pts, viewdirs, L_xyz=10, L_dir=4, netchunk=65536):
    """run_network_NeRFW, typ='coarse', training (models/nerfw.py:47-60): xyz + direction encodings into the
    `output_transient=False` branch -> [R,N,4] = [rgb_s(3), sigma_s]."""
    R, N = pts.shape[:2]
    flat = pts.reshape(-1, 3)
    dirs = viewdirs[:, None].expand(pts.shape).reshape(-1, 3)
    out = []
    for i in range(0, flat.shape[0], netchunk):
        sl = slice(i, i + netchunk)
        out.append(nerfh_static(p, posenc(flat[sl], L_xyz), posenc(dirs[sl], L_dir)))
    return torch.cat(out, 0).reshape(R, N, 4)


def composite_coarse_train(raw4, z, noise=None, white_bkgd=False):
    """raw2outputs_NeRFW, typ='coarse', test_time=False, output_transient=False (models/rendering.py:150-193,231-243;
    quirk Q1: render_rays passes white_bkgd into the output_transient slot, False here): alpha = 1-exp(-delta*relu(sigma
    + noise)), rgb = sum w c, depth = sum w z, disp = 1/max(1e-10, depth / sum w).  `noise` = randn * raw_noise_std."""
    rgb_s, sig = raw4[..., :3], raw4[..., 3]
    s = sig if noise is None else sig + noise
    alpha = 1 - torch.exp(-_deltas(z) * torch.relu(s))
    w = alpha * _excl_cumprod(1 - alpha)
    acc = w.sum(-1)
    rgb = (w[..., None] * rgb_s).sum(-2)
    if white_bkgd:
        rgb = rgb + (1 - acc[:, None])
    depth = (w * z).sum(-1)
    disp = 1. / torch.max(1e-10 * torch.ones_like(depth), depth / w.sum(-1))
    return dict(rgb=rgb, disp=disp, acc=acc, weights=w, depth=depth)


def stratified_z(z, t_rand):
    """Stratified jitter of the coarse depths (models/rendering.py:277-285): one draw per interval between midpoints."""
    mids = .5 * (z[..., 1:] + z[..., :-1])
    upper = torch.cat([mids, z[..., -1:]], -1)
    lower = torch.cat([z[..., :1], mids], -1)
    return lower + (upper - lower) * t_rand


def render_rays_train(rows, coarse, fine, emb_a, emb_t, Nc, Ni, t_rand=None, noise=None, u=None, perturb=1.,
                      raw_noise_std=0., netchunk=65536, stages=None, lindisp=False):
    """Training-mode render of packed ray rows (models/rendering.py:245-337 with test_time=False, white_bkgd=False).  The three random draws of the reference are inputs so that results are reproducible:
    t_rand [R,Nc] = torch.rand (stratified jitter, only if perturb > 0), noise [R,Nc] = torch.randn (x raw_noise_std,
    coarse alpha only), u [R,Ni] = torch.rand (importance sampling, only if perturb > 0; linspace otherwise).  Returns the
    reference's dict: rgb_map, disp_map, acc_map, raw, rgb0, disp0, acc0, z_std, transient_sigmas, beta."""
    o, d = rows[:, 0:3], rows[:, 3:6]
    near, far = rows[:, 6:7], rows[:, 7:8]
    view, hist = rows[:, 8:11], rows[:, 11:]
    R = rows.shape[0]
    z = coarse_z(near, far, Nc, R, lindisp)
    if perturb > 0.:
        if t_rand is None:
            t_rand = torch.rand(R, Nc)
        z = stratified_z(z, t_rand)
    pts = o[:, None, :] + d[:, None, :] * z[..., None]
    raw_c = query_coarse_static(coarse, pts, view, netchunk=netchunk)
    if noise is None:
        noise = torch.randn(R, Nc)   # the reference draws even when raw_noise_std = 0 (quirk Q4)
    c = composite_coarse_train(raw_c, z, noise * raw_noise_std)
    mid = .5 * (z[:, 1:] + z[:, :-1])
    if perturb > 0. and u is None:
        u = torch.rand(R, Ni)
    zs = sample_pdf(mid, c["weights"][:, 1:-1], Ni, det=(perturb == 0.), u=u if perturb > 0. else None).detach()
    zf, _ = torch.sort(torch.cat([z, zs], -1), -1)
    pts_f = o[:, None, :] + d[:, None, :] * zf[..., None]
    raw = query_fine(fine, emb_a, emb_t, pts_f, view, hist, netchunk=netchunk)
    f = composite_fine(raw, zf, test_time=False)
    if stages is not None:
        stages.update(z_coarse=z, raw_coarse=raw_c, weights_coarse=c["weights"], z_samples=zs, z_fine=zf, raw=raw,
                      weights_fine=f["weights"], depth_fine=f["depth"])
    return dict(rgb_map=f["rgb"], disp_map=f["disp"], acc_map=f["acc"], raw=raw, rgb0=c["rgb"], disp0=c["disp"], acc0=c["acc"],
                z_std=torch.std(zs, dim=-1, unbiased=False), transient_sigmas=f["transient_sigmas"], beta=f["beta"])


def nerfw_loss(results, targets, coef=1., lambda_u=0.01):
    """NerfWLoss (models/losses.py:19-57): c_l coarse colour, f_l fine colour weighted by 1/(2 beta^2), b_l = 3 + mean log
    beta, s_l = lambda_u * mean transient sigma.  `results` keys as run_nerf.py:54-58."""
    ret = {'c_l': 0.5 * ((results['rgb_coarse'] - targets) ** 2).mean(),
           'f_l': ((results['rgb_fine'] - targets) ** 2 / (2 * results['beta'].unsqueeze(1) ** 2)).mean(),
           'b_l': 3 + torch.log(results['beta']).mean(),
           's_l': lambda_u * results['transient_sigmas'].mean()}
    return {k: coef * v for k, v in ret.items()}


def train_step(rows, target, coarse, fine, emb_a, emb_t, Nc, Ni, t_rand, noise, u, perturb=1., raw_noise_std=0.):
    """One optimisation step's forward + backward (run_nerf.py:50-66): render(retraw=True, **render_kwargs_train) ->
    NerfWLoss -> sum -> backward.  Returns (loss dict, psnr, {name: gradient}) with names 'coarse.<key>', 'fine.<key>',
    'embedding_a.weight', 'embedding_t.weight'; parameters the loss does not reach are absent (None in torch)."""
    P = {"coarse." + k: v.detach().clone().requires_grad_(True) for k, v in coarse.items()}
    P.update({"fine." + k: v.detach().clone().requires_grad_(True) for k, v in fine.items()})
    P["embedding_a.weight"] = emb_a.detach().clone().requires_grad_(True)
    P["embedding_t.weight"] = emb_t.detach().clone().requires_grad_(True)
    c = {k[7:]: v for k, v in P.items() if k.startswith("coarse.")}
    f = {k[5:]: v for k, v in P.items() if k.startswith("fine.")}
    out = render_rays_train(rows, c, f, P["embedding_a.weight"], P["embedding_t.weight"], Nc, Ni, t_rand, noise, u, perturb,
                            raw_noise_std)
    ld = nerfw_loss({'rgb_fine': out['rgb_map'], 'rgb_coarse': out['rgb0'], 'beta': out['beta'],
                     'transient_sigmas': out['transient_sigmas']}, target)
    loss = sum(ld.values())
    loss.backward()
    with torch.no_grad():
        mse = ((out['rgb_map'] - target) ** 2).mean()
        ps = -10. * torch.log(mse) / math.log(10.)
    return {k: v.detach() for k, v in ld.items()}, ps, {k: v.grad for k, v in P.items() if v.grad is not None}, \
        {k: v.detach() for k, v in out.items()}


def train_step_grad_rays(rays_o, rays_d, near, far, img_idx, target, coarse, fine, emb_a, emb_t, Nc, Ni, t_rand, noise, u, perturb=1.,
                         raw_noise_std=0.):
    """d sum(NerfWLoss) / d (rays_o, rays_d) of the TRAINING render by autograd (models/rendering.py:245-337 with test_time=False
    under loss.backward(): pts = o + d z enter both networks, viewdirs = d / |d| both direction encodings; z carries none —
    z_samples.detach()).  Returns (g_o, g_d) [R,3]."""
    o = rays_o.detach().clone().requires_grad_(True)
    d = rays_d.detach().clone().requires_grad_(True)
    out = render_rays_train(pack_ray_rows(o, d, near, far, img_idx), coarse, fine, emb_a, emb_t, Nc, Ni, t_rand, noise, u, perturb,
                            raw_noise_std)
    ld = nerfw_loss({'rgb_fine': out['rgb_map'], 'rgb_coarse': out['rgb0'], 'beta': out['beta'],
                     'transient_sigmas': out['transient_sigmas']}, target)
    sum(ld.values()).backward()
    return o.grad, d.grad


def render_grad_rays(rays_o, rays_d, G, coarse, fine, emb_a, emb_t, Nc, Ni, near, far, img_idx):
    """d sum(rgb * G) / d (rays_o, rays_d) by autograd through render(rays=...) — viewdirs are derived from
    rays_d inside render (rendering.py:366-371), so their normalisation is part of the gradient."""
    o = rays_o.detach().clone().requires_grad_(True)
    d = rays_d.detach().clone().requires_grad_(True)
    rgb = render(0, 0, 0., 1 << 30, coarse, fine, emb_a, emb_t, Nc, Ni, near, far, img_idx, rays=(o, d))[0]
    (rgb * G).sum().backward()
    return rgb.detach(), o.grad, d.grad


def render_grad_c2w(H, W, focal, c2w, G, coarse, fine, emb_a, emb_t, Nc, Ni, near, far, img_idx):
    """d sum(rgb * G) / d c2w[:3,:4] by autograd through render(c2w=...) (direct_feature_matching.py:340-376)."""
    p = torch.as_tensor(c2w, dtype=torch.get_default_dtype())[:3, :4].detach().clone().requires_grad_(True)   # (float64 under tests/yardstick.py)
    rgb = render(H, W, focal, 1 << 30, coarse, fine, emb_a, emb_t, Nc, Ni, near, far, img_idx, c2w=p)[0]
    (rgb * G).sum().backward()
    return rgb.detach(), p.grad


def psnr(rgb, gt):
    """-10 log10(mean((rgb-gt)^2)) (models/rendering.py:431-433)."""
    return -10. * math.log10(float(((rgb - gt) ** 2).mean()))
