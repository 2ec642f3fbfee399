"""MI355X-native GATsSPG 2D-3D matcher (OnePose hot path) -- see DESIGN.md."""
from .gats_superglue import GATsSuperGlue, GATsSPGEngine, KeypointEncoder  # noqa: F401

__all__ = ["GATsSuperGlue", "GATsSPGEngine", "KeypointEncoder"]
