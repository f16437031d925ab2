"""Shared host-side plumbing of the operator Functions."""
import torch


def require_gpu(name, *tensors):
    """The reference's CPU branches raise NameError before reaching C (they never allocate `output`,
    FilterInterpolationLayer.py:23,32; FlowProjectionLayer.py:21-22,32); say so instead."""
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError(
                "%s: CPU tensors are not supported -- there is no CPU path (the reference's own CPU "
                "branch is unrunnable); move the tensors to the GPU" % name)


def check(err, name):
    """The reference prints a non-zero error code and carries on with zero-filled results
    (FlowProjectionLayer.py:33-34); a drop-in that silently returns zeros hides real bugs, so raise."""
    if err != 0:
        raise RuntimeError("%s returned %d: shape/stride check failed or the launch failed" % (name, err))


def f32c(t):
    """contiguous float32 view/copy (the reference caches `.contiguous()` copies for backward,
    FilterInterpolationLayer.py:14-16, and only ever sees torch.cuda.FloatTensor)."""
    if t.dtype != torch.float32:
        raise TypeError("expected a float32 tensor, got %s" % t.dtype)
    return t.contiguous()
