"""MI355X-native GATsSPG 2D-3D matcher (OnePose hot path) and the SuperPoint extractor in front of it -- see DESIGN.md."""
from .runtime import configure_hip_queues, StreamRing  # noqa: F401

configure_hip_queues()   # one hardware queue per frame in flight, unless the caller exported GPU_MAX_HW_QUEUES (runtime.py)

from .gats_superglue import GATsSuperGlue, GATsSPGEngine, KeypointEncoder  # noqa: F401
from .superpoint import SuperPoint, SuperPointEngine  # noqa: F401
from .frame_matcher import FrameMatcher  # noqa: F401

__all__ = ["GATsSuperGlue", "GATsSPGEngine", "KeypointEncoder", "SuperPoint", "SuperPointEngine", "FrameMatcher", "StreamRing", "configure_hip_queues"]
