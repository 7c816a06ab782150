"""Wire format of the query keypoint cloud (`sensor_msgs/PointCloud2.data` inside
`gisnav_msgs/OrthoStereoImage.query_sift`).

Mirrors `KEYPOINT_DTYPE` (ros/gisnav/gisnav/core/_shared.py:26-35), produced by TwistNode
(core/twist_node.py:175-202) and consumed by PoseNode with `np.frombuffer`
(core/pose_node.py:207-213).  532 bytes per keypoint, little endian:
x, y, z, size, angle (5 x f32) followed by the 128-float SIFT descriptor.  The PointCloud2
`fields` metadata the producer attaches is wrong (descriptor declared at offset 12) and is
ignored by the consumer, so it is ignored here too.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import numpy as np

KEYPOINT_DTYPE = np.dtype(
    [
        ("x", np.float32),
        ("y", np.float32),
        ("z", np.float32),
        ("size", np.float32),
        ("angle", np.float32),
        ("descriptor", np.float32, (128,)),
    ]
)


def pack_keypoints(xy: np.ndarray, size: np.ndarray, angle: np.ndarray, desc: np.ndarray) -> bytes:
    """What TwistNode._publish_keypoints serialises (twist_node.py:175-187)."""
    n = len(xy)
    rec = np.empty(n, dtype=KEYPOINT_DTYPE)
    rec["x"], rec["y"], rec["z"] = xy[:, 0], xy[:, 1], 0.0
    rec["size"], rec["angle"], rec["descriptor"] = size, angle, desc
    return rec.tobytes()


def unpack_keypoints(data: bytes):
    """pose_node.py:207-213 -> (kp (N,2) f32, desc (N,128) f32, size (N,), angle (N,))."""
    rec = np.frombuffer(data, dtype=KEYPOINT_DTYPE)
    return np.column_stack((rec["x"], rec["y"])), rec["descriptor"], rec["size"], rec["angle"]


@dataclass
class Stamp:
    sec: int = 0
    nanosec: int = 0


@dataclass
class ImageMsg:
    """The subset of sensor_msgs/Image PoseNode touches (mono8 rasters)."""
    data: np.ndarray
    stamp: Stamp = field(default_factory=Stamp)


@dataclass
class CameraInfo:
    """The subset of sensor_msgs/CameraInfo PoseNode touches: flat row-major 3x3 `k`."""
    k: np.ndarray
    height: int = 0
    width: int = 0


@dataclass
class OrthoStereoImage:
    """ros/gisnav_msgs/msg/OrthoStereoImage.msg:14-18 without the ROS envelope."""
    query_sift: bytes
    reference: ImageMsg
    dem: ImageMsg
    crs: str = ""
    query_stamp: Optional[Stamp] = None
