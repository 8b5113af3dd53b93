"""Shared argument checks of the extension shims: the pybind modules raise RuntimeError (C++ exceptions: AT_ASSERTM /
TV_ASSERT_RT_ERR / TORCH_CHECK) on non-contiguous or misplaced tensors."""
import functools

import torch

from .._lib import Df3dError


def runtime_errors(fn):
    """Library errors (Df3dError is a RuntimeError already) and argument errors come out as RuntimeError, like pybind's."""
    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        try:
            return fn(*args, **kwargs)
        except (ValueError, AssertionError, IndexError) as e:
            raise RuntimeError("%s: %s" % (fn.__name__, e))
    return wrapped


def need_cuda_contiguous(t, name):
    if not isinstance(t, torch.Tensor):
        raise RuntimeError("%s must be a tensor" % name)
    if not t.is_cuda:
        raise Df3dError("%s must be a CUDA tensor (the MI355X build has no CPU path)" % name)
    if not t.is_contiguous():
        raise RuntimeError("%s tensor has to be contiguous" % name)
    return t
