"""gisnav_amd/ros2_node.py: the optional rclpy wrapper.  Without ROS 2 (this image) the module imports and refuses to build the node class with a
clear message; its message conversion is plain Python on duck-typed messages and is checked here against the wire dataclasses the GPU shim takes."""
import types

import numpy as np
import pytest

from gisnav_amd import ros2_node as rn
from gisnav_amd.wire import KEYPOINT_DTYPE, pack_keypoints


def _img(arr, sec, nanosec, step_pad=0, encoding="mono8"):
    h, w = arr.shape
    step = w + step_pad
    buf = np.zeros((h, step), np.uint8)
    buf[:, :w] = arr
    return types.SimpleNamespace(height=h, width=w, step=step, encoding=encoding, data=buf.tobytes(),
                                 header=types.SimpleNamespace(stamp=types.SimpleNamespace(sec=sec, nanosec=nanosec)))


def test_module_imports_without_ros_and_refuses_to_build_the_node():
    with pytest.raises(ImportError, match="ROS 2 environment"):
        rn.make_node_class()


def test_ortho_stereo_image_conversion_matches_the_wire_dataclass():
    rng = np.random.default_rng(3)
    n = 7
    kp = rng.uniform(0, 100, (n, 2)).astype(np.float32)
    raw = pack_keypoints(kp, rng.uniform(1, 5, n).astype(np.float32), rng.uniform(0, 360, n).astype(np.float32), rng.uniform(0, 255, (n, 128)).astype(np.float32))
    assert len(raw) == n * KEYPOINT_DTYPE.itemsize == n * 532
    ref = rng.integers(0, 255, (12, 20), dtype=np.uint8)
    dem = rng.integers(0, 255, (12, 20), dtype=np.uint8)
    msg = types.SimpleNamespace(
        query=types.SimpleNamespace(header=types.SimpleNamespace(stamp=types.SimpleNamespace(sec=0, nanosec=0))),      # empty query image: the stamp comes from the keypoints (pose_node.py:489-495)
        query_sift=types.SimpleNamespace(data=np.frombuffer(raw, np.uint8), header=types.SimpleNamespace(stamp=types.SimpleNamespace(sec=41, nanosec=7))),
        reference=_img(ref, 5, 6, step_pad=4), dem=_img(dem, 5, 6), crs=types.SimpleNamespace(data="+proj=affine +xoff=1"))
    w = rn.ortho_stereo_image_from_ros(msg)
    assert w.query_sift == raw and (w.query_stamp.sec, w.query_stamp.nanosec) == (41, 7)
    assert np.array_equal(w.reference.data, ref) and np.array_equal(w.dem.data, dem)            # the row padding of `step` is dropped
    assert (w.reference.stamp.sec, w.reference.stamp.nanosec) == (5, 6) and w.crs == "+proj=affine +xoff=1"
    msg.query.header.stamp.sec = 99
    assert rn.ortho_stereo_image_from_ros(msg).query_stamp.sec == 99                              # a stamped query image wins
    with pytest.raises(ValueError):
        rn.image_to_mono8(_img(ref, 0, 0, encoding="bgr8"))


def test_camera_info_conversion():
    k = [205.5, 0, 320, 0, 205.5, 240, 0, 0, 1]
    c = rn.camera_info_from_ros(types.SimpleNamespace(k=k, height=480, width=640))
    assert c.k.shape == (9,) and c.k.dtype == np.float64 and np.allclose(c.k, k) and (c.height, c.width) == (480, 640)
