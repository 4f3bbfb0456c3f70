"""craft_amd — MI355X-native implementation of CRAFT's inner-loop hot path.

``craft_amd.CRAFT`` is a drop-in for the reference's ``core.network.CRAFT``; the hot path
(correlation volume + lookup, SETrans attention, SepConvGRU refinement) runs on hand-written
gfx950 HIP kernels behind the C ABI of ``include/craft_hip.h`` (``libcraft_hip.so``).
"""
from .network import CRAFT, GraphedForward  # noqa: F401
from .utils import InputPadder, default_args, load_checkpoint  # noqa: F401

RAFTER = CRAFT  # alias the reference keeps for un-pickling old checkpoints (evaluate.py:18-19)
