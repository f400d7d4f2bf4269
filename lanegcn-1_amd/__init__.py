"""MI355X-native LaneGCN graph-convolution hot path (MapNet LaneConv + A2M/M2M/M2A/A2A).

Host side mirrors the reference's ``lanegcn.py`` / ``layers.py`` / ``utils.py`` / ``data.py``
interfaces; all arithmetic of the hot path runs in hand-written HIP kernels (``csrc/``) reached
through the C ABI declared in ``include/lgcn.h`` (``liblgcn.so``, loaded with ctypes).
There is no CPU fallback: importing ``lanegcn_amd.ops`` without the built library raises.
"""
__version__ = "0.1.0"
