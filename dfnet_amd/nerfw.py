"""Host-side mirror of /root/reference/script/models/nerfw.py for the HIP render path.

`NeRFW` here is a *parameter container* with the reference's exact state_dict layout (so the
reference's `{:06d}.tar` checkpoints load unchanged) and the reference's initialisation order
(constructor reseeds the global RNG to 0, nerfw.py:245, then builds the Linear layers in the same
sequence, so a no-checkpoint run starts from the same weights).  The arithmetic is not here:
`create_nerf` packs the weights into a `NerfHEngine` (MFMA fragment layout on the GPU) and the
returned render kwargs route `rendering.render` to it.
"""
import os

import torch
import torch.nn as nn

from .engine import NerfHEngine
from . import optim

img2mse = lambda x, y: torch.mean((x - y) ** 2)
mse2psnr = lambda x: -10. * torch.log(x) / torch.log(torch.Tensor([10.]))


def to8b(x):
    """uint8(255 * clip(x, 0, 1)) — truncation, not rounding (reference models/nerf.py:11)."""
    import numpy as np
    return (255 * np.clip(x, 0, 1)).astype(np.uint8)


def _act(layer, act):
    return nn.Sequential(layer, act) if act is not None else layer


class NeRFW(nn.Module):
    """Weights of one NeRF-H network ('coarse' or 'fine'); names as nerfw.py:259-295."""

    def __init__(self, typ, D=8, W=256, skips=[4], in_channels_xyz=63, in_channels_dir=27,
                 encode_appearance=False, in_channels_a=48, encode_transient=False, in_channels_t=16,
                 beta_min=0.1, out_ch_size=3):
        super().__init__()
        torch.manual_seed(0)  # reference side effect (nerfw.py:245), part of the contract (quirk Q12)
        if out_ch_size != 3:
            raise NotImplementedError("feature-rendering heads (out_ch_size != 3) are not part of the NeRF-H path")
        self.typ, self.D, self.W, self.skips = typ, D, W, list(skips)
        self.in_channels_xyz, self.in_channels_dir = in_channels_xyz, in_channels_dir
        self.encode_appearance = False if typ == 'coarse' else encode_appearance
        self.in_channels_a = in_channels_a if encode_appearance else 0
        self.encode_transient = False if typ == 'coarse' else encode_transient
        self.in_channels_t = in_channels_t
        self.beta_min = beta_min
        for i in range(D):
            k = in_channels_xyz if i == 0 else (W + in_channels_xyz if i in skips else W)
            setattr(self, f"xyz_encoding_{i + 1}", _act(nn.Linear(k, W), nn.ReLU(True)))
        self.xyz_encoding_final = nn.Linear(W, W)
        self.dir_encoding = _act(nn.Linear(W + in_channels_dir + self.in_channels_a, W // 2), nn.ReLU(True))
        self.static_sigma = _act(nn.Linear(W, 1), nn.Softplus())
        self.static_rgb = _act(nn.Linear(W // 2, 3), nn.Sigmoid())
        if self.encode_transient:
            self.transient_encoding = nn.Sequential(
                nn.Linear(W + in_channels_t, W // 2), nn.ReLU(True), nn.Linear(W // 2, W // 2), nn.ReLU(True),
                nn.Linear(W // 2, W // 2), nn.ReLU(True), nn.Linear(W // 2, W // 2), nn.ReLU(True))
            self.transient_sigma = _act(nn.Linear(W // 2, 1), nn.Softplus())
            self.transient_rgb = _act(nn.Linear(W // 2, 3), nn.Sigmoid())
            self.transient_beta = _act(nn.Linear(W // 2, 1), nn.Softplus())

    def forward(self, *a, **k):
        raise RuntimeError("NeRFW here only holds weights; evaluate it through dfnet_amd.rendering.render "
                           "(the fused HIP kernels take rays, not pre-embedded vectors)")


class HipQuery:
    """Stands in for the reference's `network_query_fn` lambda (nerfw.py:425-434): carries the engine
    that evaluates both networks.  Calling it with pre-sampled points evaluates the MLP stage."""

    def __init__(self, engine, netchunk=65536, trainer=None, modules=None):
        self.engine = engine
        self.netchunk = netchunk
        self.trainer = trainer      # NerfHTrainer: the training-mode render and its gradients (nerf_train.py)
        self.modules = modules      # (network_fn, network_fine, embedding_a, embedding_t) behind the engine's packed weights
        self.stale = False          # the master weights moved (optimizer step) since the engine was packed
        self._packed_versions = self._versions()

    def _versions(self):
        """torch's in-place version counters of every master tensor: they move with optimizer.step() / load_state_dict()."""
        if self.modules is None:
            return None
        return tuple(p._version for m in self.modules for p in m.parameters())

    def refresh(self):
        """Re-pack the test-time engine from the (trained) modules — before a validation render — when they moved: either the
        training loop said so (`stale`) or their version counters did."""
        if self.modules is None:
            return
        cur = self._versions()
        if self.stale or cur != self._packed_versions:
            self.engine.load_modules(*self.modules)
            self.stale = False
            self._packed_versions = cur

    def __call__(self, *a, **k):
        raise RuntimeError("network_query_fn is fused into the HIP render path; call rendering.render()")


def create_nerf(args):
    """Same contract as nerfw.py:356-502: returns (render_kwargs_train, render_kwargs_test, start,
    grad_vars, optimizer).  Differences: the networks live in a NerfHEngine (`network_query_fn.engine`)."""
    if not getattr(args, "NeRFH", False):
        raise NotImplementedError("only the NeRF-H path (--NeRFH) is implemented")
    if not getattr(args, "encode_hist", False):
        raise ValueError("NeRF-H needs --encode_hist (the reference leaves embedding_a/t undefined without it, nerfw.py:385-391)")
    if args.reduce_embedding != -1 or args.i_embed != 0:
        raise NotImplementedError("only the paper-default positional encoding (reduce_embedding=-1, i_embed=0) is implemented")
    if not args.use_viewdirs:
        raise NotImplementedError("use_viewdirs=False is not part of the NeRF-H path")
    if not torch.cuda.is_available():
        raise RuntimeError("create_nerf needs a GPU: the render path is HIP-only, there is no CPU fallback")
    device = torch.device("cuda", torch.cuda.current_device())
    input_ch, input_ch_views = 3 + 6 * args.multires, 3 + 6 * args.multires_views
    dim_a, rem_a = divmod(args.in_channels_a, args.hist_bin)
    dim_t, rem_t = divmod(args.in_channels_t, args.hist_bin)
    if rem_a or rem_t or (dim_a, dim_t) != (5, 2):
        raise ValueError("in_channels_a/t must equal hist_bin*5 / hist_bin*2 (nn.Embedding(N_vocab,5)/(N_vocab,2), nerfw.py:386-390)")
    embedding_a = nn.Embedding(args.N_vocab, 5).to(device)
    embedding_t = nn.Embedding(args.N_vocab, 2).to(device)
    model = NeRFW('coarse', D=args.netdepth, W=args.netwidth, skips=[4], in_channels_xyz=input_ch,
                  in_channels_dir=input_ch_views).to(device)
    grad_vars = list(model.parameters())
    model_fine = None
    if args.N_importance > 0:
        model_fine = NeRFW('fine', D=args.netdepth, W=args.netwidth, skips=[4], in_channels_xyz=input_ch,
                           in_channels_dir=input_ch_views, encode_appearance=True, encode_transient=True,
                           in_channels_a=args.in_channels_a, in_channels_t=args.in_channels_t).to(device)
        grad_vars += list(model_fine.parameters()) + list(embedding_a.parameters()) + list(embedding_t.parameters())
    else:
        raise NotImplementedError("N_importance == 0 (no fine network) is not part of the NeRF-H path")

    if args.no_grad_update:
        grad_vars, optimizer = None, None
    else:
        optimizer = optim.Adam(params=grad_vars, lr=args.lrate, betas=(0.9, 0.999))   # torch.optim.Adam, one launch per step

    start = 0
    if args.ft_path is not None and args.ft_path != 'None':
        ckpts = [args.ft_path]
    else:
        d = os.path.join(args.basedir, args.expname)
        ckpts = [os.path.join(d, f) for f in sorted(os.listdir(d)) if 'tar' in f] if os.path.isdir(d) else []
    print('Found ckpts', ckpts)
    if len(ckpts) > 0 and not args.no_reload:
        print('Reloading from', ckpts[-1])
        ckpt = torch.load(ckpts[-1], map_location=device)
        start = ckpt['global_step']
        model.load_state_dict(ckpt['network_fn_state_dict'])
        model_fine.load_state_dict(ckpt['network_fine_state_dict'])
        embedding_a.load_state_dict(ckpt['embedding_a_state_dict'])
        embedding_t.load_state_dict(ckpt['embedding_t_state_dict'])

    engine = NerfHEngine(depth=args.netdepth, width=args.netwidth, multires=args.multires,
                         multires_views=args.multires_views, hist_bin=args.hist_bin, dim_a=dim_a, dim_t=dim_t,
                         n_vocab=args.N_vocab, precision=getattr(args, "precision", "f16x3"))
    engine.load_modules(model, model_fine, embedding_a, embedding_t)
    if engine.fast and getattr(args, "precision", "f16x3") == "f16x3" and getattr(args, "coarse_precision", "same") == "f16":
        engine.set_render_options(coarse_f16=True)   # opt-in: the split-f16 fine network makes the pixel, an f16 coarse network places its samples

    from .nerf_train import NerfHTrainer
    query = HipQuery(engine, args.netchunk, modules=(model, model_fine, embedding_a, embedding_t))
    if not args.no_grad_update:
        query.trainer = NerfHTrainer(engine, model, model_fine, embedding_a, embedding_t)
    render_kwargs_train = {
        'network_query_fn': query, 'perturb': args.perturb,
        'N_importance': args.N_importance, 'network_fine': model_fine, 'N_samples': args.N_samples,
        'network_fn': model, 'use_viewdirs': args.use_viewdirs, 'white_bkgd': args.white_bkgd,
        'raw_noise_std': args.raw_noise_std, 'embedding_a': embedding_a, 'embedding_t': embedding_t,
        'test_time': False,
    }
    if args.dataset_type != 'llff' or args.no_ndc:
        print('Not ndc!')
        render_kwargs_train['ndc'] = False
        render_kwargs_train['lindisp'] = args.lindisp
    render_kwargs_test = dict(render_kwargs_train)
    render_kwargs_test.update(perturb=False, raw_noise_std=0., test_time=True)
    return render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer
