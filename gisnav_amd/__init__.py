"""gisnav_amd -- MI355X-native (gfx950) implementation of GISNav's PoseNode hot path.

Camera-frame <-> map-tile pose estimation: SIFT-descriptor LightGlue matching + dual-softmax
mutual-NN match head + PnP/RANSAC, behind the reference's three seams
(ros/gisnav/gisnav/core/pose_node.py:109-121,285-287; core/_shared.py:89-125).
Hand-written HIP kernels behind a C ABI (include/gisnav_amd.h); Python host code only marshals.
"""
__version__ = "0.1.0"

__all__ = ["LightGlueMatcher", "PoseEngine", "PoseNode", "compute_pose"]


def __getattr__(name):  # lazy: importing the package must not need a GPU or the .so
    if name == "LightGlueMatcher":
        from .matcher import LightGlueMatcher
        return LightGlueMatcher
    if name == "PoseEngine":
        from .engine import PoseEngine
        return PoseEngine
    if name == "PoseNode":
        from .pose_node import PoseNode
        return PoseNode
    if name == "compute_pose":
        from .pose import compute_pose
        return compute_pose
    raise AttributeError(name)
