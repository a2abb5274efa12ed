"""Step-dependent scalars for HIP-graph replay.

A training step recorded with `torch.cuda.graph` freezes every by-value kernel argument.  What changes from step to step on this path is
tiny - the optimiser's bias corrections / learning rate (11 floats) and the sampler's pcg32 state (two 64-bit words) - so the recorded
kernels read it from DEVICE memory (arcn_adam_ema_step_replay, arcn_march_count_replay) and the host rewrites that memory before each
replay: one small asynchronous copy from a ring of pinned slots.  The host runs ahead of the device, so a slot must not be rewritten
before the copy that reads it has run: the ring has 1024 slots and the CALLER keeps the host within that many pushes of the device
(trainer.GraphedTrainStep holds it within 8 steps by watching its sample totals arrive in pinned memory).  The copy is a KERNEL
reading the pinned slot (arcn_copy_words): hipMemcpyAsync operations between graph launches stall the stream on this stack once the host
runs a few dozen operations ahead (tools/exp_graph_module.py: 4.9 ms per step instead of 0.75)."""
import numpy as np
import torch


def copy_words(src, dst):
    """dst <- src (same byte size, a multiple of 4; device or pinned-host tensors) as a KERNEL on the current stream: hipMemcpyAsync
    operations queued between graph launches stall the stream once the host runs ahead (arcn_copy_words, csrc/optim.hip)"""
    from .. import _native as N
    n = src.numel() * src.element_size()
    assert n == dst.numel() * dst.element_size() and n % 4 == 0 and src.is_contiguous() and dst.is_contiguous()
    N.check(N.lib().arcn_copy_words(src.data_ptr(), dst.data_ptr(), n // 4, N.stream()), 'copy_words')


class ReplayScalars:
    def __init__(self, device, n_bytes, slots=1024):
        self.device = torch.device(device)
        self.n_bytes = int(n_bytes)
        assert self.n_bytes % 4 == 0
        self.dev = torch.zeros(self.n_bytes, dtype=torch.uint8, device=self.device)
        self._pinned = torch.zeros((slots, self.n_bytes), dtype=torch.uint8).pin_memory()
        self._np = self._pinned.numpy()
        self._k = 0

    def push(self, payload):
        """payload: bytes / numpy array of n_bytes -> queued on the current stream into `self.dev`"""
        k = self._k
        self._k = (k + 1) % self._np.shape[0]
        self._np[k, :] = np.frombuffer(bytes(payload), dtype=np.uint8) if not isinstance(payload, np.ndarray) else payload.view(np.uint8).reshape(-1)
        copy_words(self._pinned[k], self.dev)
