"""The float64 yardstick of the gradient tests.

Several gradients on this path are ill-conditioned in fp32 whatever computes them: autograd of torch.cumprod divides by the factors
1 - alpha (which vanish on opaque samples), conv1_1's weight gradient is a cancelling sum over 307 200 pixels, a ReLU or max-pool tie
within round-off moves a whole gradient element.  A tolerance for such a tensor is MEASURED, not argued: the same computation through
the CPU oracle in float64 is the truth, the distance of torch's own fp32 autograd (the oracle in fp32, or the reference's recorded
output) from it is the yardstick, and the HIP result is held to a small multiple of that yardstick."""
import contextlib

import torch


@contextlib.contextmanager
def float64_default():
    """Run the oracle in float64: its constructors (linspace, ones, zeros ...) follow torch's default dtype."""
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        yield
    finally:
        torch.set_default_dtype(prev)


def to64(x):
    if isinstance(x, dict):
        return {k: to64(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(to64(v) for v in x)
    if torch.is_tensor(x) and x.is_floating_point():
        return x.detach().cpu().double()
    return x


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def rays_off_a_gate(got, truth64, single64, max_frac=0.04, trials=6, shift=3e-6, seed=0):
    """Boolean mask of the rays a batch criterion may be applied to.

    These maps are piecewise smooth: a ray one of whose hidden units sits within round-off of a ReLU gate has a DISCONTINUOUS gradient
    there, and any fp32 implementation lands on one side or the other (measured, tools/gpu_debug_raygrad.py: the float64 gradient of
    the worst ray of the netwidth-32 test moves by 6.2e-2 under a 2e-6 shift of its origin — exactly the HIP gradient's distance from
    it).  So for the few rays with the largest error (at most max_frac of the batch) the float64 gradient is re-evaluated at `trials`
    origins shifted by ~`shift`: a ray whose float64 gradient itself moves by at least half the observed error is on a gate and
    is left out; every other ray stays in.  got / truth64: [R, k] per-ray gradients; single64(i, delta3) -> float64 gradient [k] of
    ray i with its origin shifted by delta3."""
    got, truth64 = got.detach().cpu().double(), truth64.detach().cpu().double()
    R = got.shape[0]
    err = (got - truth64).norm(dim=1)
    rel = err / truth64.norm(dim=1).clamp_min(1e-30)
    keep = torch.ones(R, dtype=torch.bool)
    gen = torch.Generator().manual_seed(seed)
    for i in err.argsort(descending=True)[: max(1, int(max_frac * R))].tolist():
        if float(rel[i]) < 1e-3:
            break
        moved = 0.0
        for _ in range(trials):
            delta = torch.randn(3, dtype=torch.float64, generator=gen) * shift
            g = single64(i, delta).detach().double().reshape(-1)
            moved = max(moved, float((g - truth64[i]).norm() / truth64[i].norm().clamp_min(1e-30)))
        if moved >= 0.5 * float(rel[i]):
            keep[i] = False
    return keep


def gradient_on_a_gate(grad64, observed_rel, trials=3, scales=(3e-7, 1e-6, 3e-6), seed=0):
    """Is an observed gradient error of size `observed_rel` (relative L2 of the worst tensor) the signature of a ReLU / max-pool tie within
    the forward's round-off?  A network of ReLUs and max-pools is piecewise linear in its inputs: where a pre-activation (or the gap
    between the two largest entries of a pooling window) is below round-off, the parameter gradient is DISCONTINUOUS, and two correct
    implementations with different summation orders land on different sides (measured, tools/gpu_debug_params.py, seed 32: every wrong
    element of conv4_3's weight gradient sits in ONE output channel — one flipped gate at one pixel — while its bias gradient and all
    layers above are at 5e-7).  That is a property of the point, not of the implementation, and it is checkable: `grad64(rel_scale,
    generator)` evaluates the float64 gradient {name: tensor} with inputs AND weights multiplied by (1 + rel_scale * U(-1, 1)) and
    `grad64(0, None)` is the unperturbed truth.  The scales stay at or below the forward's own parity tolerance (3e-6 against the 5e-6
    relative L2 the features are held to): a perturbation no larger than the arithmetic's documented error.  Returns (on_gate, moved,
    scale): moved = the largest relative-L2 change of any tensor; on_gate = moved >= half the observed error.  An error that the
    float64 gradient's own round-off sensitivity does not explain is a bug."""
    base = grad64(0.0, None)
    gen = torch.Generator().manual_seed(seed)
    moved = 0.0
    for scale in scales:
        for _ in range(trials):
            g = grad64(scale, gen)
            for k in base:
                moved = max(moved, rel_l2(g[k], base[k]))
            if moved >= 0.5 * observed_rel:
                return True, moved, scale
    return False, moved, scales[-1]
