"""Small device-side copies as KERNELS.

`copy_words` moves a few bytes between device and / or pinned host memory with a kernel on the current stream instead of a runtime
copy operation: hipMemcpyAsync operations queued between the launches of a step stall the stream on this stack once the host runs a few
dozen operations ahead (4.9 ms per step instead of 0.75).  trainer.FusedNgpStep sends each step's sample total to pinned
memory this way and reads it a step later."""


def copy_words(src, dst):
    """dst <- src (same byte size, a multiple of 4; device or pinned-host tensors) as a KERNEL on the current stream (arcn_copy_words,
    csrc/optim.hip)"""
    from .. import _native as N
    n = src.numel() * src.element_size()
    assert n == dst.numel() * dst.element_size() and n % 4 == 0 and src.is_contiguous() and dst.is_contiguous()
    N.check(N.lib().arcn_copy_words(src.data_ptr(), dst.data_ptr(), n // 4, N.stream()), 'copy_words')
