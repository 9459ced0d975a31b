"""MI355X-native implementation of the MARS mask-proposal alignment-and-ranking path.

The directory name is not a Python identifier; load it with
``__graft_entry__.load_package()`` (or ``importlib.import_module`` on the directory
name with the repository root on ``sys.path``).  Sub-modules:

  synth      seeded synthetic weights / episodes (CPU, torch only)
  hip        ctypes binding of ``csrc/libmars_hip.so`` (the C-ABI; raises if missing)
  ops        tensor-level wrappers over the C-ABI (device pointers + stream)
  models     DINOv2 / CLIP / AlphaCLIP encoders with the reference's method names
  mars       MARS facade and its components (reference API surface)
  utils      BackboneLoader seam
"""
from . import synth, tokenizer  # noqa: F401  (pure host code, always importable)

__all__ = ["synth"]


def _load_native():
    """Import everything that needs the compiled library; raises if it is missing."""
    from . import hip  # noqa: F401
    hip.lib()  # raises a clear error when libmars_hip.so has not been built
    from . import ops, models, mars, utils, matcher, segment_anything  # noqa: F401
    from .mars import ClipPipeline  # noqa: F401
    return hip
