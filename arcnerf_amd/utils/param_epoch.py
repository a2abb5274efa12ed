"""A process-wide counter of "parameters were rewritten behind torch's back".

The optimiser kernels (FusedAdam, the scatter-fused Adam of NgpPipeline.train_step) update parameters through raw device pointers: torch's
per-tensor version counters do not move.  Anything that caches a function of the parameters (ops.sdf_chain.padded_params: the weight-normed,
padded weights of the wide sdf net, built once per step instead of once per pass) keys its cache on `current()` as well as on the tensors'
own versions; every raw writer calls `bump()`."""
_EPOCH = 0


def bump():
    global _EPOCH
    _EPOCH += 1
    return _EPOCH


def current():
    return _EPOCH
