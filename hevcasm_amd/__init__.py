"""hevcasm_amd -- MI355X (gfx950) native implementation of the HEVCasm inner-loop kernel library.

The product is the C-ABI shared library ``lib/libhevcasm_hip.so`` (headers in ``include/``):
the reference's ``hevcasm_*`` tables / selectors / self-tests plus the batched
``hevcasm_hip_*_batch`` entry points that run hand-written HIP kernels.  This Python package is
plumbing around it for tests and benchmarks:

* ``capi``      ctypes signatures of every exported entry point; loading fails loudly if the
                library has not been built (``python -c "import __graft_entry__ as g; g.build()"``).
* ``batch``     thin wrappers taking torch CUDA tensors (device memory, streams).
* ``workload``  seeded synthetic frame stacks and block lists.
* ``shard``     contiguous split of a batch across ranks (one process per GPU, no collectives).

There is no CPU fallback anywhere in this package: the batch entry points return an error code
when no gfx950 device is usable and ``batch`` raises on it.
"""
from . import capi  # noqa: F401

__all__ = ["capi"]
