"""MI355X-native MoCoGAN training hot path.

``csrc/``      hand-written HIP kernels for gfx950 + the C ABI declared in ``include/mocogan_hip.h``
``hiplib.py``  ctypes binding of ``lib/libmocogan_hip.so`` (raw device pointers, current HIP stream)
``layout.py``  reference (Chainer) layout <-> device layout of activations and parameters
``nets.py``    ImageGenerator / ImageDiscriminator / VideoDiscriminator on those kernels
``step.py``    one training iteration (losses, hand-scheduled backward, Adam) + data-parallel exchange
The reference-facing API (``model.net``, ``model.updater``, ``train.py``) at the repo root is a
thin layer over these modules.
"""
from .build import build, lib_path  # noqa: F401
