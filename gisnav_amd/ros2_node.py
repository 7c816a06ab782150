"""Optional rclpy glue around seam B3: a ROS 2 node with the reference PoseNode's topics whose pose comes from `gisnav_amd.pose_node.PoseNode`.

The reference node (ros/gisnav/gisnav/core/pose_node.py) subscribes to `/camera/camera_info` (constants.py:69) and to StereoNode's
`/gisnav/stereo_node/pose_image` (OrthoStereoImage; pose_node.py:499-505).  From the solver's (r, t) it forms the pose of `camera_optical` in the
REP 105 `earth` frame (pose_node.py:333-381) and then walks the tf tree (gisnav_map <- earth, camera_link_optical -> base_link: pose_node.py:383-470)
to publish `~/pose` in `gisnav_map`.  This module is the MINIMAL node: same two subscriptions, and the EARTH-frame camera pose on `~/pose_earth`
(PoseWithCovarianceStamped, frame_id "earth") -- everything up to pose_node.py:381.  The tf2 buffer / broadcasters, the debug images
(pose_node.py:310-347) and the EKF `set_pose` client (pose_node.py:136-163) are the ROS control plane and stay with the reference: a deployment
that wants the complete node applies INTEGRATION.md 2's two-line edit to the reference's own PoseNode, which keeps all of it.  What runs here:

    ROS messages -> `gisnav_amd.wire` dataclasses (the bytes of `query_sift.data` untouched, the mono8 rasters as numpy views)
                 -> `PoseNode.estimate` (matcher + DEM lift + PnP on the GPU: pose_node.py:207-308)
                 -> `georef.pose_to_earth` (pose_node.py:333-381: camera position through the CRS affine to WGS 84 to ECEF, orientation ENU -> ECEF)
                 -> PoseWithCovarianceStamped, stamped like pose_node.py:489-495

rclpy, sensor_msgs, geometry_msgs and gisnav_msgs are NOT part of this repository's image: importing this module always works, building the node
class needs them (`make_node_class()` raises ImportError naming what is missing).  The message conversion is plain Python on duck-typed
messages and is tested on the CPU (tests/test_ros2_glue.py)."""
from __future__ import annotations

from typing import Any, Optional

import numpy as np

from .wire import CameraInfo, ImageMsg, OrthoStereoImage, Stamp

ROS_TOPIC_CAMERA_INFO = "/camera/camera_info"                 # constants.py:69
ROS_TOPIC_POSE_IMAGE = "/gisnav/stereo_node/pose_image"       # f"/{ROS_NAMESPACE}/" + ROS_TOPIC_RELATIVE_POSE_IMAGE.replace("~", STEREO_NODE_NAME): pose_node.py:500-503
ROS_TOPIC_POSE_EARTH = "~/pose_earth"                         # (the reference's `~/pose`, constants.py:64, is the tf-chained gisnav_map pose: not published here)


def _stamp(header: Any) -> Stamp:
    s = header.stamp
    return Stamp(int(s.sec), int(s.nanosec))


def image_to_mono8(img: Any) -> np.ndarray:
    """sensor_msgs/Image (mono8) -> (height, width) uint8 view of its data, honouring `step` (what `CvBridge.imgmsg_to_cv2(msg, "mono8")` returns for
    a mono8 image: pose_node.py:216, 221-223).  Other encodings are refused: StereoNode publishes mono8 rasters (stereo_node.py:262-275)."""
    if getattr(img, "encoding", "mono8") not in ("mono8", "8UC1"):
        raise ValueError(f"expected a mono8 raster, got encoding {img.encoding!r}")
    h, w = int(img.height), int(img.width)
    step = int(getattr(img, "step", w)) or w
    buf = np.frombuffer(bytes(img.data) if not isinstance(img.data, (bytes, bytearray, memoryview, np.ndarray)) else img.data, dtype=np.uint8)
    if buf.size < h * step:
        raise ValueError("image data shorter than height x step")
    return buf[: h * step].reshape(h, step)[:, :w]


def ortho_stereo_image_from_ros(msg: Any) -> OrthoStereoImage:
    """gisnav_msgs/OrthoStereoImage -> the wire dataclass `PoseNode.estimate` takes.  `query_sift.data` is passed on as bytes: the 532-byte
    KEYPOINT_DTYPE records go to the device as they are."""
    data = msg.query_sift.data
    raw = data.tobytes() if hasattr(data, "tobytes") else bytes(data)
    q_stamp = _stamp(msg.query.header)
    if q_stamp.sec == 0:                                       # pose_node.py:489-495: an empty query image carries no stamp, the keypoint cloud does
        q_stamp = _stamp(msg.query_sift.header)
    return OrthoStereoImage(query_sift=raw,
                            reference=ImageMsg(image_to_mono8(msg.reference), _stamp(msg.reference.header)),
                            dem=ImageMsg(image_to_mono8(msg.dem), _stamp(msg.dem.header)),
                            crs=str(msg.crs.data), query_stamp=q_stamp)


def camera_info_from_ros(msg: Any) -> CameraInfo:
    return CameraInfo(k=np.asarray(msg.k, np.float64).reshape(9), height=int(msg.height), width=int(msg.width))


def pose_fields(r: np.ndarray, t: np.ndarray, crs: str, ref_shape) -> Optional[dict]:
    """(r, t) of the solver -> the fields of the published pose: ECEF position, (x, y, z, w) orientation (pose_node.py:333-381); None when the
    camera centre falls outside the reference raster (pose_node.py:341-343)."""
    from .georef import pose_to_earth
    return pose_to_earth(r, t, crs, ref_shape)


def make_node_class():
    """Build the rclpy Node subclass (import-guarded: rclpy and the message packages are absent from this repository's image)."""
    try:
        import rclpy  # noqa: F401
        from rclpy.node import Node
        from rclpy.qos import QoSPresetProfiles
        from geometry_msgs.msg import PoseWithCovarianceStamped
        from sensor_msgs.msg import CameraInfo as RosCameraInfo
        from gisnav_msgs.msg import OrthoStereoImage as RosOrthoStereoImage
    except ImportError as exc:  # pragma: no cover - exercised by tests through the message below
        raise ImportError("gisnav_amd.ros2_node needs a ROS 2 environment (rclpy, sensor_msgs, geometry_msgs, gisnav_msgs): " + str(exc)) from exc

    from .pose_node import PoseNode

    class GisnavAmdPoseNode(Node):
        """`~/pose_earth` from `/camera/camera_info` + StereoNode's pose image, computed by gisnav_amd (one message = one GPU call, ~1 ms)."""

        def __init__(self, state_dict, *args, device: int = 0, precision: str = "f16x2_f16_attn", **kwargs):
            super().__init__(*args, **kwargs)
            self._impl = PoseNode(state_dict, device=device, precision=precision)
            self._camera_info: Optional[CameraInfo] = None
            qos = QoSPresetProfiles.SENSOR_DATA.value
            self._pub = self.create_publisher(PoseWithCovarianceStamped, ROS_TOPIC_POSE_EARTH, qos)
            self.create_subscription(RosCameraInfo, ROS_TOPIC_CAMERA_INFO, self._camera_info_cb, qos)
            self.create_subscription(RosOrthoStereoImage, ROS_TOPIC_POSE_IMAGE, self._pose_image_cb, qos)

        def _camera_info_cb(self, msg) -> None:
            self._camera_info = camera_info_from_ros(msg)

        def _pose_image_cb(self, msg) -> None:
            if self._camera_info is None:                      # narrow_types: no result until both inputs exist (_decorators.py:117-160)
                return
            # one malformed message (a non-mono8 raster, a short keypoint buffer, more keypoints than the context can grow to) must not kill the node:
            # exceptions of the conversion and of the estimator are logged and the message is dropped
            try:
                wire = ortho_stereo_image_from_ros(msg)
                pose = self._impl.estimate(self._camera_info, wire)
            except Exception as exc:  # noqa: BLE001
                self.get_logger().warning(f"pose_image message dropped: {exc.__class__.__name__}: {exc}")
                return
            if pose is None:
                self.get_logger().warning(f"no pose ({self._impl.last_num_matches} matches)")
                return
            fields = pose_fields(pose[0], pose[1], wire.crs, wire.reference.data.shape)
            if fields is None:
                self.get_logger().warning("camera centre outside the reference raster - no pose")
                return
            out = PoseWithCovarianceStamped()      # covariance stays all-zero, as in the reference (pose_node.py:478-495 fills pose and header only)
            out.header.frame_id = "earth"
            out.header.stamp.sec, out.header.stamp.nanosec = wire.query_stamp.sec, wire.query_stamp.nanosec
            p, q = fields["position"], fields["orientation"]
            out.pose.pose.position.x, out.pose.pose.position.y, out.pose.pose.position.z = float(p[0]), float(p[1]), float(p[2])
            o = out.pose.pose.orientation
            o.x, o.y, o.z, o.w = float(q[0]), float(q[1]), float(q[2]), float(q[3])
            self._pub.publish(out)

    return GisnavAmdPoseNode
