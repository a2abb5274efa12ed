"""EMA (arcnerf/trainer/ema.py:6-43): after every optimiser step each trainable parameter becomes the de-biased running average
    new = ((1 - d) p + d old (1 - d^(n-1))) / (1 - d^n)
and the average is written BACK into the parameter.  With optim.FusedAdam(ema_decay=...) that arithmetic already ran inside the
optimiser kernel (arcn_adam_ema_step): this class then only keeps the reference's interface (`set_n_step`, `ema_step`) and forwards
the step count; with any other optimiser it applies the update itself with torch foreach ops on the device."""
import torch

from ..optim import FusedAdam


class EMA(object):
    def __init__(self, model, decay, optimizer=None):
        self.model = model
        self.decay = decay
        self.n_step = 0
        self.fused = isinstance(optimizer, FusedAdam) and optimizer.ema_decay is not None
        self.optimizer = optimizer
        if self.fused:
            if abs(float(optimizer.ema_decay) - float(decay)) > 0:
                raise ValueError('EMA decay {} differs from the FusedAdam ema_decay {}'.format(decay, optimizer.ema_decay))
            self.old_avg = None
        else:
            self.old_avg = self.get_model_params()

    def get_model_params(self):
        return {n: p.detach().clone() for n, p in self.model.named_parameters() if p.requires_grad}

    def set_n_step(self, n_step):
        self.n_step = n_step
        if self.fused:
            self.optimizer.set_ema_n_step(n_step)

    def ema_step(self):
        self.n_step += 1
        if self.fused:       # applied by the optimiser's kernel in the step that just ran
            if self.optimizer.ema_n_step is not None and self.optimizer.ema_n_step != self.n_step:
                raise RuntimeError('EMA.ema_step() must follow every FusedAdam.step() (EMA at {}, optimiser at {})'.format(
                    self.n_step, self.optimizer.ema_n_step))
            return
        d = self.decay
        deb_old = 1 - d ** (self.n_step - 1)
        deb_new = 1.0 / (1 - d ** self.n_step)
        with torch.no_grad():
            names, ps = zip(*[(n, p) for n, p in self.model.named_parameters() if p.requires_grad])
            olds = [self.old_avg[n] for n in names]
            new = torch._foreach_mul(list(ps), 1 - d)
            torch._foreach_add_(new, olds, alpha=d * deb_old)
            torch._foreach_mul_(new, deb_new)
            for p, o, v in zip(ps, olds, new):
                p.copy_(v)
                o.copy_(v)
