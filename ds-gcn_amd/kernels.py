"""Fused ops of the DS-GCN hot path, each a ``torch.autograd.Function`` over the C-ABI HIP library.

PyTorch is plumbing here (device memory, the current stream, autograd bookkeeping); every byte of
arithmetic in these ops happens in ``libdsgcn.so`` (``ds-gcn_amd/csrc``).  There is NO CPU or
eager fallback: inputs must be CUDA fp32 tensors and the library must load, otherwise the op raises.

``ops()`` returns the active op namespace (this module).  ``use_ops(ns)`` is a test seam that swaps
in another namespace with the same signatures (``tests/torch_ops.py``) so the host-side wiring can be
checked against the oracle without a GPU; the product never calls it.
"""
import contextlib
import sys

_active = None


def ops():
    return _active if _active is not None else sys.modules[__name__]


@contextlib.contextmanager
def use_ops(namespace):
    global _active
    prev = _active
    _active = namespace
    try:
        yield namespace
    finally:
        _active = prev


NAME = 'hip'
