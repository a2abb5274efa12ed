"""The losses of the training steps on the path (arcnerf/loss/img_loss.py:11-100 ImgLoss + HuberLoss, arcnerf/loss/geo_loss.py:12-70 EikonalLoss,
arcnerf/loss/__init__.py:41-66 AllLoss):
`loss['sum'] = weight * mean(loss_fn(output[key], data['img']))` summed over the configured keys.  Note the reference's Huber is
0.5 / delta * d^2 inside the band and |d| - 0.5 delta outside - torch.nn.functional.huber_loss DIVIDED by delta."""
import torch
import torch.nn as nn

from ..utils.cfgs_utils import get_value_from_cfgs_field


class HuberLoss(nn.Module):
    def __init__(self, delta=1.0, reduction='none'):
        super().__init__()
        self.delta, self.reduction = delta, reduction

    def forward(self, x, y):
        a = (x - y).abs()
        loss = torch.where(a < self.delta, 0.5 / self.delta * (a ** 2), a - 0.5 * self.delta)
        return loss.mean() if self.reduction == 'mean' else loss


class ImgLoss(nn.Module):
    def __init__(self, cfgs=None):
        super().__init__()
        self.keys = get_value_from_cfgs_field(cfgs, 'keys', ['rgb'])
        t = get_value_from_cfgs_field(cfgs, 'loss_type', 'MSE')
        if t == 'MSE':
            self.loss = nn.MSELoss(reduction='none')
        elif t == 'L1':
            self.loss = nn.L1Loss(reduction='none')
        elif t == 'Huber':
            self.loss = HuberLoss(get_value_from_cfgs_field(cfgs, 'delta', 1.0), reduction='none')
        else:
            raise NotImplementedError('Loss type {} not support in img loss...'.format(t))
        self.internal_weights = get_value_from_cfgs_field(cfgs, 'internal_weights', None)
        self.use_mask = get_value_from_cfgs_field(cfgs, 'use_mask', False)
        self.do_mean = get_value_from_cfgs_field(cfgs, 'do_mean', True)

    def forward(self, data, output):
        gt = data['img'].to(output[self.keys[0]].device)
        if (isinstance(self.loss, HuberLoss) and self.do_mean and not self.use_mask and self.internal_weights is None and gt.is_cuda
                and gt.dtype == torch.float32 and all(output[k].is_cuda and output[k].dtype == torch.float32 and output[k].shape == gt.shape
                                                      for k in self.keys)):
            # the NGP recipe (Huber, plain mean): value and gradient from one kernel per key
            from ..ops.autograd import HuberMeanFn
            total = None
            for k in self.keys:
                v = HuberMeanFn.apply(output[k], gt, float(self.loss.delta))
                total = v if total is None else total + v
            return total
        loss = 0.0
        for i, k in enumerate(self.keys):
            w = 1.0 if self.internal_weights is None else self.internal_weights[i]
            loss = loss + w * self.loss(output[k], gt)
        if self.do_mean:
            if self.use_mask:
                mask = data['mask'].to(gt.device)
                per = loss.mean(-1)          # mean_tensor_by_mask (common/utils/torch_utils.py:223-247): per batch row, then the mean
                dims = tuple(range(1, per.dim()))
                loss = ((per * mask).sum(dim=dims) / (mask.sum(dim=dims) + 1e-5)).mean()
            else:
                loss = loss.mean()
        return loss


class EikonalLoss(nn.Module):
    """arcnerf/loss/geo_loss.py:12-70: loss_type(|n|, 1) over `key` ('normal' (B, N, 3) or 'normal_pts' (B, N, P, 3)), mean (by mask if
    `use_mask`)"""

    def __init__(self, cfgs=None):
        super().__init__()
        self.key = get_value_from_cfgs_field(cfgs, 'key', 'normal')
        t = get_value_from_cfgs_field(cfgs, 'loss_type', 'MSE')
        if t == 'MSE':
            self.loss = nn.MSELoss(reduction='none')
        elif t == 'L1':
            self.loss = nn.L1Loss(reduction='none')
        else:
            raise NotImplementedError('Loss type {} not support in geo loss...'.format(t))
        self.use_mask = get_value_from_cfgs_field(cfgs, 'use_mask', False)
        self.do_mean = get_value_from_cfgs_field(cfgs, 'do_mean', True)

    def forward(self, data, output):
        out = output[self.key]
        norm = torch.norm(out, dim=-1)
        loss = self.loss(norm, torch.ones_like(norm))
        if self.do_mean:
            if self.use_mask:
                mask = data['mask'].to(out.device)
                if loss.dim() == 3:
                    mask = torch.repeat_interleave(mask.unsqueeze(-1), loss.shape[-1], -1)
                dims = tuple(range(1, loss.dim()))
                loss = ((loss * mask).sum(dim=dims) / (mask.sum(dim=dims) + 1e-5)).mean()       # mean_tensor_by_mask (torch_utils.py:223-247)
            else:
                loss = loss.mean()
        return loss


class AllLoss(object):
    """build_loss(cfgs): every entry of cfgs.loss, multiplied by its `weight`; returns {'sum', 'names', <name>: value}"""

    REGISTRY = {'ImgLoss': ImgLoss, 'EikonalLoss': EikonalLoss}

    def __init__(self, loss_cfgs):
        self.funcs, self.names, self.weights = [], [], []
        for name, c in loss_cfgs.__dict__.items():
            if name not in self.REGISTRY:
                raise NotImplementedError('loss {} is outside the hot path (DESIGN.md 9)'.format(name))
            self.funcs.append(self.REGISTRY[name](c))
            self.names.append(name)
            self.weights.append(c.weight)

    def __call__(self, inputs, output):
        loss = {'sum': 0.0, 'names': []}
        for f, n, w in zip(self.funcs, self.names, self.weights):
            loss[n] = f(inputs, output) * w
            loss['sum'] = loss['sum'] + loss[n]
            loss['names'].append(n)
        return loss


def build_loss(cfgs, logger=None):
    return AllLoss(cfgs.loss)
