"""torch.optim-shaped front end of the fused Adam (+ EMA) kernel.

The reference builds `torch.optim.Adam` in common/trainer/optimizer.py and applies `EMA.ema_step` afterwards
(arcnerf/trainer/ema.py:29-43: a debiased running average written back into the parameter).  torch's
multi-tensor Adam streams the 12.6 M NGP parameters through about ten kernels per step; `arcn_adam_ema_step` does Adam, the EMA
and the gradient clear in one pass (488 MB of HBM traffic, ~75 us; 390 MB with `ema_in_param`).  FusedAdam is a drop-in for torch.optim.Adam on CUDA float32
parameters: same update rule (L2 weight decay added to the gradient, bias correction, eps outside the square root), same
`state_dict` layout (`step`, `exp_avg`, `exp_avg_sq`), plus an optional fused EMA and zero-on-step."""
import torch

from .ops import functional as F


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, ema_decay=None, zero_grad_on_step=False,
                 ema_in_param=False):
        """ema_decay: None = plain Adam; a float also applies the reference's EMA.ema_step in the same pass (debiased running
        average of the parameter, stored in state['ema'] AND written back into the parameter, arcnerf/trainer/ema.py:29-43).
        zero_grad_on_step: the kernel clears .grad while it reads it (then `optimizer.zero_grad()` can be skipped).
        ema_in_param: the written-back average makes state['ema'] equal to the parameter after every step; when nothing else writes
        the parameters between steps (no clipping of weights, no reload mid-run) the shadow can be the parameter itself: no
        state['ema'], one 8 B/param sweep less, same bits."""
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1:
            raise ValueError('invalid Adam hyper-parameter')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.ema_decay = ema_decay
        self.zero_grad_on_step = zero_grad_on_step
        self.ema_in_param = bool(ema_in_param and ema_decay is not None)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()
                        and p.data_ptr() % 16 == 0 and p.grad.data_ptr() % 16 == 0):
                    raise RuntimeError('FusedAdam handles contiguous, 16-byte aligned float32 CUDA parameters (there is no CPU fallback)')
                st = self.state[p]
                if len(st) == 0:
                    st['step'] = 0
                    st['exp_avg'] = torch.zeros_like(p)
                    st['exp_avg_sq'] = torch.zeros_like(p)
                if self.ema_decay is not None and not self.ema_in_param and st.get('ema') is None:
                    # also after load_state_dict of a checkpoint written by torch.optim.Adam (step / exp_avg / exp_avg_sq only)
                    st['ema'] = p.detach().clone()
                st['step'] = int(st['step']) + 1
                F.adam_ema_step(p, p.grad, st['exp_avg'], st['exp_avg_sq'], p if self.ema_in_param else st.get('ema'), st['step'], lr=group['lr'],
                                betas=group['betas'], eps=group['eps'], weight_decay=group['weight_decay'],
                                ema_decay=self.ema_decay if self.ema_decay is not None else 0.0, zero_grad=self.zero_grad_on_step)
        return loss
