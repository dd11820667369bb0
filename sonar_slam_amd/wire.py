"""Feature-cloud wire format between the feature node and the SLAM node (host, numpy only).

``publish_features`` (bruce_slam/src/bruce_slam/feature_extraction.py:175-193) ships the N x 2
feature cloud (y_forward, x_lateral) as a ``PointCloud2`` built by ``create_cloud_xyz32``: three
little-endian float32 per point, (x, y, z) = (forward, 0, lateral), point_step 12.
``SLAMNode.SLAM_callback`` (slam_ros.py:169-170) reads it back as ``[x, -z]``.  Skipped frames carry
one NaN point (feature_extraction.py:205-206, slam_ros.py:173-174).  These helpers produce / parse
exactly that byte payload so a replay (or a rospy wrapper) stays byte-compatible with the other
consumers of the topic (mapping node, rviz).
"""
import numpy as np

POINT_STEP = 12
FIELDS = (("x", 0), ("y", 4), ("z", 8))   # sensor_msgs/PointField FLOAT32, count 1


def pack_features(points):
    """N x 2 (forward, lateral) -> PointCloud2.data bytes (feature_extraction.py:182-185)."""
    points = np.asarray(points)
    xyz = np.c_[points[:, 0], np.zeros(len(points)), points[:, 1]]
    return np.ascontiguousarray(xyz, "<f4").tobytes()


def unpack_features(data, remove_nans=False):
    """PointCloud2.data bytes -> the N x 2 cloud the SLAM node works with (slam_ros.py:169-170):
    ``ros_numpy.point_cloud2.pointcloud2_to_xyz_array(msg)`` then ``np.c_[x, -1 * z]``.  ros_numpy (un-vendored)
    builds that array with ``get_xyz_points(..., dtype=np.float)``: a FLOAT64 array holding the float32 values of the
    message -- which is why ``Keyframe.transform_points`` computes in double on the SLAM side -- so this returns float64
    too.  ``remove_nans=True`` additionally drops non-finite points like ros_numpy's default does (then the one-NaN
    message of a skipped frame arrives EMPTY and slam_ros.py:173 never fires: the reference's own quirk); the default
    keeps them so that ``is_skipped`` can do what slam_ros.py:173 intends."""
    xyz = np.frombuffer(data, "<f4").reshape(-1, 3).astype(np.float64)
    if remove_nans:
        xyz = xyz[np.isfinite(xyz).all(axis=1)]
    return np.c_[xyz[:, 0], -1 * xyz[:, 2]]


def is_skipped(points):
    """slam_ros.py:173: a frame whose first point is NaN carries no features."""
    return bool(len(points) and np.isnan(points[0, 0]))
