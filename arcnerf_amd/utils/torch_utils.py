"""chunk_processing: the chunk plug-in API of the reference (common/utils/torch_utils.py:79-220).

Slices every tensor / ndarray / dict-of-arrays argument on dim 0 in `chunk_size` steps, calls `func` per chunk and
concatenates the per-chunk outputs (tensors, arrays, dicts of them; anything else is collected into lists).
chunk_size <= 0 or no array argument => one direct call.  Outputs are gathered per field and concatenated ONCE
(the reference re-concatenates after every chunk, an O(n^2) copy pattern).
"""
import contextlib

import numpy as np
import torch


def is_torch_or_np(x):
    return isinstance(x, (torch.Tensor, np.ndarray))


def torch_to_np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else t


def np_wrapper(func, *args):
    """call a torch function with numpy arguments (first dim batch), return numpy"""
    conv = [torch.tensor(a, dtype=torch.float32) if isinstance(a, np.ndarray) else a for a in args]
    out = func(*conv)
    if isinstance(out, tuple):
        return tuple(torch_to_np(o) for o in out)
    return torch_to_np(out)


def _batch_size(args):
    n = 0
    for a in args:
        vals = a.values() if isinstance(a, dict) else [a]
        for v in vals:
            if is_torch_or_np(v):
                assert n == 0 or n == v.shape[0], 'Batch size for array not matched...'
                n = v.shape[0]
    return n


def _slice(a, lo, hi, to_gpu):
    moved = False

    def one(v):
        nonlocal moved
        if not is_torch_or_np(v):
            return v
        v = v[lo:hi]
        if to_gpu and isinstance(v, torch.Tensor) and not v.is_cuda:
            v = v.cuda(non_blocking=True)
            moved = True
        return v

    out = {k: one(v) for k, v in a.items()} if isinstance(a, dict) else one(a)
    return out, moved


def _cat(parts):
    first = parts[0]
    if len(parts) == 1 and isinstance(first, (torch.Tensor, np.ndarray)):
        return first     # one chunk: its outputs are the result (the reference's torch.cat of one tensor is a copy of it)
    if isinstance(first, torch.Tensor):
        return torch.cat(parts, dim=0)
    if isinstance(first, np.ndarray):
        return np.concatenate(parts, axis=0)
    return list(parts)


def chunk_processing(func, chunk_size, gpu_on_func, *args):
    if chunk_size <= 0:
        return func(*args)
    n = _batch_size(args)
    if n == 0:
        return func(*args)
    fields = None
    # (the networks are the same objects in every iteration: their split / padded weights are made once, ops.functional.split_weight_scope)
    from ..ops.functional import split_weight_scope
    with split_weight_scope() if n > chunk_size else contextlib.nullcontext():
        for lo in range(0, n, chunk_size):
            moved = False
            sliced = []
            for a in args:
                s, m = _slice(a, lo, lo + chunk_size, gpu_on_func)
                moved |= m
                sliced.append(s)
            out = func(*sliced)
            out = list(out) if isinstance(out, (tuple, list)) else [out]
            if moved:
                out = [{k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in o.items()} if isinstance(o, dict)
                       else (o.cpu() if isinstance(o, torch.Tensor) else o) for o in out]
            if fields is None:
                fields = [({k: [] for k in o} if isinstance(o, dict) else []) for o in out]
            for acc, o in zip(fields, out):
                if isinstance(o, dict):
                    for k, v in o.items():
                        acc[k].append(v)
                else:
                    acc.append(o)
    if fields is None:
        return None
    merged = [({k: _cat(v) for k, v in f.items()} if isinstance(f, dict) else _cat(f)) for f in fields]
    return merged[0] if len(merged) == 1 else tuple(merged)
