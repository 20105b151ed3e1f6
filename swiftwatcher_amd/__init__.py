"""MI355X-native implementation of swiftwatcher's per-frame segment(+classify) hot path.

Drop-in surface (mirrors the reference's module names):
    swiftwatcher_amd.data_structures.FrameQueue / Frame / Segment
    swiftwatcher_amd.image_filtering.<reference function names>
    swiftwatcher_amd.segment_classification.SegmentClassifier
backed by hand-written HIP kernels for gfx950 in libswk.so (C ABI: include/swk.h).
There is no CPU fallback anywhere in this package.
"""
from . import _lib                                   # noqa: F401
from ._lib import Context, SwkError, default_params  # noqa: F401

__all__ = ["Context", "SwkError", "default_params"]
