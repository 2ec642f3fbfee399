"""MI355X-native GATsSPG 2D-3D matcher (OnePose hot path) and the SuperPoint extractor in front of it -- see DESIGN.md."""
from .gats_superglue import GATsSuperGlue, GATsSPGEngine, KeypointEncoder  # noqa: F401
from .superpoint import SuperPoint, SuperPointEngine  # noqa: F401
from .frame_matcher import FrameMatcher  # noqa: F401

__all__ = ["GATsSuperGlue", "GATsSPGEngine", "KeypointEncoder", "SuperPoint", "SuperPointEngine", "FrameMatcher"]
