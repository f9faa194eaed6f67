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
