"""Mirror of /root/reference/script/models/ray_utils.py:5-15 on the HIP raygen kernel."""
import torch

from . import engine as _engine


def get_rays(H, W, focal, c2w):
    """rays_o, rays_d of an H x W pinhole image from a [3,4] (or [4,4]) c2w; both [H, W, 3] on the GPU.
    Pixel centres at integer coordinates, rays_d un-normalised (ray_utils.py:6-14)."""
    dev = torch.device("cuda", torch.cuda.current_device())
    c2w = torch.as_tensor(c2w, dtype=torch.float32, device=dev)
    o, d, _ = _engine.raygen(int(H), int(W), float(focal), c2w, want_viewdirs=False)
    return o, d


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """Forward-facing rays in normalised device coordinates (ray_utils.py:27-46); same shapes as the inputs."""
    dev = torch.device("cuda", torch.cuda.current_device())
    o = torch.as_tensor(rays_o, dtype=torch.float32, device=dev)
    d = torch.as_tensor(rays_d, dtype=torch.float32, device=dev)
    no, nd = _engine.ndc_rays(H, W, focal, near, o.reshape(-1, 3), d.reshape(-1, 3))
    return no.reshape(o.shape), nd.reshape(d.shape)
