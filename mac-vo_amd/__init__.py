"""macvo_amd — MI355X (gfx950) native hot path of MAC-VO.

Layout
  csrc/            hand-written HIP kernels + the C ABI (``include/macvo_hip.h``) -> ``libmacvo_hip.so``
  _lib.py          ctypes binding of the C ABI (fails loudly when the library is missing)
  ops.py           thin torch-tensor wrappers (device pointers + current HIP stream -> C ABI)
  interfaces.py    mirror of the reference's plugin interfaces (IFrontend, IKeypointSelector,
                   ICovariance2to3, IOptimizer, SubclassRegistry, ConfigTestable)
  plugins.py       drop-in plugin classes (HIP_CovAwareSelector(_NoDepth), HIP_MappingPointSelector,
                   HIP_MatchCovariance, HIP_TwoFrame_PGO, ...)
  pipeline.py      the per-frame hot path in the reference's call order (Odometry/MACVO.py:173-311)
  distributed.py   one process per GPU, sequence sharding, RCCL pose gather

PyTorch is used for device memory, streams and torch.distributed only; every arithmetic step of the hot
path runs in the HIP kernels.  Nothing here imports ``oracle/`` (test infrastructure).
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401  (does not load the .so until first use)
