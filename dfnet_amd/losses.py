"""Host-side mirror of /root/reference/script/models/losses.py: `ColorLoss`, `NerfWLoss`, `loss_dict`.

The tensors these see are per-ray maps ([N,3], [N]) and the [N, Nf] transient densities — tiny next to the render —
so the expressions are torch tensor ops on the device, exactly as the reference writes them; autograd hands their
gradients to the HIP render node (dfnet_amd/nerf_train.py: _RenderTrainFn).  The fused training step
(NerfHTrainer.train_step) uses the one-kernel form of NerfWLoss instead (dfn_nerfw_loss), checked against this class.
"""
import torch
from torch import nn


class ColorLoss(nn.Module):
    """losses.py:5-16."""

    def __init__(self, coef=1):
        super().__init__()
        self.coef = coef
        self.loss = nn.MSELoss(reduction='mean')

    def forward(self, inputs, targets):
        loss = self.loss(inputs['rgb_coarse'], targets)
        if 'rgb_fine' in inputs:
            loss = loss + self.loss(inputs['rgb_fine'], targets)
        return self.coef * loss


class NerfWLoss(nn.Module):
    """Equation 13 of NeRF-W as the reference implements it (losses.py:19-57): c_l coarse colour, f_l fine colour weighted by
    1 / (2 beta^2), b_l = 3 + mean(log beta), s_l = lambda_u * mean(transient sigma); every term times coef."""

    def __init__(self, coef=1, lambda_u=0.01):
        super().__init__()
        self.coef = coef
        self.lambda_u = lambda_u

    def forward(self, inputs, targets, use_hier_rgbs=False, rgb_h=None, rgb_w=None):
        ret = {'c_l': 0.5 * ((inputs['rgb_coarse'] - targets) ** 2).mean()}
        if 'rgb_fine' in inputs:
            if 'beta' not in inputs:
                ret['f_l'] = 0.5 * ((inputs['rgb_fine'] - targets) ** 2).mean()
            else:
                ret['f_l'] = ((inputs['rgb_fine'] - targets) ** 2 / (2 * inputs['beta'].unsqueeze(1) ** 2)).mean()
                ret['b_l'] = 3 + torch.log(inputs['beta']).mean()
                ret['s_l'] = self.lambda_u * inputs['transient_sigmas'].mean()
        return {k: self.coef * v for k, v in ret.items()}


loss_dict = {'color': ColorLoss, 'nerfw': NerfWLoss}
