"""Measurement side of the dynamic batch size (arcnerf/models/fg_model.py:100-130): every training forward adds
max_allowance / (valid samples + 1) to a running sum; `get_dynamicbs_factor` returns the mean and resets."""
import torch


class DynamicBsMeter:
    """The reference adds `float(max_allowance) / (float(mask_pts.sum()) + 1)` to a Python float every step (fg_model.py:105-116: one
    host read per step).  Here a step's count is ONE device-to-device copy into a ring; `factor()` (every update_epoch steps) reads the
    ring back and does the reference's double arithmetic on the exact integer counts - same numbers, no per-step synchronisation."""

    RING = 4096

    def __init__(self, max_allowance):
        self.max_allowance = max_allowance
        self.measured_batch_size = 0
        self.measured_count = 0
        self._ring, self._pending = None, 0
        self._streams = set()

    def reset(self):
        self.measured_batch_size = 0
        self.measured_count = 0
        self._pending = 0

    def add(self, n_valid, stream=None):
        """n_valid: a 0-d / 1-element device tensor (e.g. NgpPipeline.n_dev) or a number.
        stream: the stream that PRODUCED the count, when that is not the current one (the sampling stream of a batch marched ahead): the
        ring is then written and, as long as every entry came that way, read back on that stream - the read at the next batch-size update
        waits for marchers that finished steps ago instead of draining the step's stream (2.9 ms of host stall every 16 steps in the
        training loop of tools/psnr_recipe.py)."""
        if self.max_allowance <= 0:
            return
        if not torch.is_tensor(n_valid):
            self.measured_batch_size += float(self.max_allowance) / (float(n_valid) + 1)
            self.measured_count += 1
            return
        if self._ring is None or self._ring.device != n_valid.device:
            self._drain()
            self._ring = torch.zeros(self.RING, dtype=torch.int64, device=n_valid.device)
        if self._pending >= self.RING:
            self._drain()
        if stream is not None:
            with torch.cuda.stream(stream):
                self._ring[self._pending].copy_(n_valid.reshape(()))
        else:
            self._ring[self._pending].copy_(n_valid.reshape(()))
        self._streams.add(stream)
        self._pending += 1
        self.measured_count += 1

    def _drain(self):
        if self._pending:
            cap = float(self.max_allowance)
            side = [st for st in self._streams if st is not None]
            if len(side) == 1 and None not in self._streams:
                with torch.cuda.stream(side[0]):
                    vals = self._ring[:self._pending].tolist()
            else:
                for st in side:      # entries from both kinds of producer: the current stream waits for the others, then reads
                    torch.cuda.current_stream().wait_stream(st)
                vals = self._ring[:self._pending].tolist()
            self.measured_batch_size += sum(cap / (float(v) + 1.0) for v in vals)
            self._pending = 0
        self._streams = set()

    def factor(self):
        """get_dynamicbs_factor (fg_model.py:117-130): mean of the measurements since the last call (1 if none), then reset"""
        self._drain()
        f = float(self.measured_batch_size) / self.measured_count if self.measured_count > 0 else 1
        self.reset()
        return f
