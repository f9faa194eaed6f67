"""dfn_pose_orthogonalize / _backward (csrc/pose_polar.hip): the reference's `svd_reg` — pose[:, :3, :3] <- U V^T of torch.svd
(feature/direct_feature_matching.py:85-92) — as the closed-form orthogonal polar factor, forward and adjoint, against torch.svd and
its autograd in fp64."""
import pytest
import torch

from dfnet_amd.feature_misc import orthogonalize_pose

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _svd_reg(pose):
    u, s, v = torch.svd(pose[:, :3, :3])
    return torch.cat([u @ v.transpose(-2, -1), pose[:, :3, 3:]], -1)


@pytest.mark.parametrize("B,spread", [(4, 0.05), (8, 0.5), (64, 1.5)])
def test_pose_orthogonalize_vs_torch_svd_and_autograd(B, spread):
    gen = torch.Generator().manual_seed(B)
    # near-rotations (what the regressor predicts), generic matrices, and — at spread 1.5 — some with det < 0 (U V^T is then a reflection)
    q, _ = torch.linalg.qr(torch.randn(B, 3, 3, generator=gen, dtype=torch.float64))
    M = q + spread * torch.randn(B, 3, 3, generator=gen, dtype=torch.float64)
    pose64 = torch.cat([M, torch.randn(B, 3, 1, generator=gen, dtype=torch.float64)], -1).float().double().requires_grad_(True)
    G = torch.randn(B, 3, 4, generator=gen, dtype=torch.float64)
    ref = _svd_reg(pose64)
    (ref * G).sum().backward()
    pose = pose64.detach().float().to(DEV).requires_grad_(True)
    out = orthogonalize_pose(pose)
    (out * G.float().to(DEV)).sum().backward()
    ef = float((out.detach().double().cpu() - ref.detach()).abs().max())
    eg = float((pose.grad.double().cpu() - pose64.grad).norm() / pose64.grad.norm())
    print(f"pose orthogonalisation B {B} spread {spread}: forward max abs {ef:.2e}, gradient rel L2 {eg:.2e}, "
          f"min det {float(torch.linalg.det(M).min()):.2f}")
    assert ef < 5e-7 and eg < 1e-5
    R = out.detach()[:, :, :3].double()
    assert float((R.transpose(-2, -1) @ R - torch.eye(3, device=DEV, dtype=torch.float64)).abs().max()) < 1e-6
    assert torch.equal(out.detach()[:, :, 3], pose.detach()[:, :, 3])


def test_pose_orthogonalize_cpu_path_and_singular_input():
    p = torch.randn(3, 3, 4)
    assert torch.allclose(orthogonalize_pose(p), _svd_reg(p), atol=1e-6)


def test_rank_deficient_rotation_blocks_give_a_finite_orthogonal_factor_like_torch_svd():
    """A zero-initialised or collapsed fc_pose output (rank 0, 1, 2 rotation blocks): torch.svd — what the reference calls — returns a
    finite U V^T there, so must this (round-5 advisor finding: NaN poisoned the DFNet_dm loss silently).  The factor is unique only up
    to the orientation of the null directions, so it is checked by its defining properties: finite, orthogonal, M = Q P with
    P = Q^T M symmetric positive semi-definite; for rank 2 the gradient is finite and equals torch.svd's autograd in fp64 up to the
    sign convention of the free direction (compared through a loss that does not see it); non-finite input stays non-finite."""
    gen = torch.Generator().manual_seed(7)
    a, b = torch.randn(3, generator=gen, dtype=torch.float64), torch.randn(3, generator=gen, dtype=torch.float64)
    blocks = {
        "rank 0": torch.zeros(3, 3, dtype=torch.float64),
        "rank 1": torch.outer(a, b),
        "rank 2": torch.outer(a, b) + torch.outer(torch.randn(3, generator=gen, dtype=torch.float64), torch.randn(3, generator=gen, dtype=torch.float64)),
        "rank 2, axis aligned": torch.diag(torch.tensor([2.0, 0.5, 0.0], dtype=torch.float64)),
        "rank 1, axis aligned": torch.diag(torch.tensor([0.0, 3.0, 0.0], dtype=torch.float64)),
    }
    for name, M in blocks.items():
        pose = torch.cat([M, torch.tensor([[0.1], [0.2], [0.3]], dtype=torch.float64)], 1)[None].float().to(DEV)
        out = orthogonalize_pose(pose)
        Q = out[0, :, :3].double().cpu()
        Mf = pose[0, :, :3].double().cpu()
        assert bool(torch.isfinite(Q).all()), name
        assert float((Q.T @ Q - torch.eye(3, dtype=torch.float64)).abs().max()) < 1e-6, name
        P = Q.T @ Mf
        scale = max(float(Mf.abs().max()), 1.0)
        assert float((P - P.T).abs().max()) < 1e-6 * scale and float(torch.linalg.eigvalsh(0.5 * (P + P.T)).min()) > -1e-6 * scale, name
        assert float((Q @ P - Mf).abs().max()) < 1e-6 * scale, name
        assert torch.equal(out[0, :, 3], pose[0, :, 3])
    # rank 2: differentiable; a loss through Q P-invariant quantities (here the first two left singular directions' image) has torch.svd's gradient
    M = blocks["rank 2"].float().double()
    p64 = torch.cat([M, torch.zeros(3, 1, dtype=torch.float64)], 1)[None].requires_grad_(True)
    u, s_, v = torch.svd(p64[:, :3, :3])
    W = (v[0, :, :2] @ v[0, :, :2].T).detach()            # projector on the row space: Q W is independent of the free direction
    ((u @ v.transpose(-2, -1))[0] @ W).pow(2).mul(torch.arange(9, dtype=torch.float64).reshape(3, 3)).sum().backward()
    pd = p64.detach().float().to(DEV).requires_grad_(True)
    (orthogonalize_pose(pd)[0, :, :3].double() @ W.to(DEV)).pow(2).mul(torch.arange(9, dtype=torch.float64, device=DEV).reshape(3, 3)).sum().backward()
    assert bool(torch.isfinite(pd.grad).all())
    assert float((pd.grad.double().cpu() - p64.grad).norm() / p64.grad.norm()) < 1e-4
    bad = torch.full((1, 3, 4), float("nan"), device=DEV)
    assert bool(torch.isnan(orthogonalize_pose(bad)[0, :, :3]).all())
