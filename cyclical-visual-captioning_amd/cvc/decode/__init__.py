"""cvc.decode -- the inference side of the hot path (reference model/captioner.py:384-443): checkpoint packing (weights.py), the
engine (engine.py) and its per-path launch lists (path_packed.py, path_tile.py, path_ring.py; path_experimental.py for the schedules
of include/cvc_hip_experimental.h)."""
from .weights import *          # noqa: F401,F403
from .weights import _segs      # noqa: F401
from .engine import DecodeEngine  # noqa: F401
