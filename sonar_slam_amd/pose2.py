"""A minimal gtsam.Pose2 for the ROS-free / gtsam-free harnesses (host, fp64).

gtsam (python, version unpinned in the reference: README.md) is absent from this image, so the
host-side pose algebra the reference does with ``gtsam.Pose2`` (compose / between / inverse /
matrix / theta) is restated here from gtsam's published Pose2/Rot2 source: the rotation is kept as
(c, s); a product is ``fromCosSin(c1*c2 - s1*s2, s1*c2 + c1*s2)`` and is renormalised only when
|c^2 + s^2 - 1| > 1e-10 (``Rot2::normalize``); ``theta() = atan2(s, c)``.  Parity unpinned.  Code
that is handed real ``gtsam.Pose2`` objects never touches this class.
"""
import math

import numpy as np


class Pose2(object):
    """Minimal gtsam.Pose2: rotation kept as (c, s) like gtsam::Rot2, products renormalised only
    when |c^2 + s^2 - 1| > 1e-10 (Rot2::normalize)."""

    __slots__ = ("_x", "_y", "_c", "_s")

    def __init__(self, x=0.0, y=0.0, theta=0.0, _cs=None):
        self._x, self._y = float(x), float(y)
        if _cs is None:
            self._c, self._s = math.cos(theta), math.sin(theta)
        else:
            c, s = _cs
            scale = c * c + s * s
            if abs(scale - 1.0) > 1e-10:
                scale = 1.0 / math.sqrt(scale)
                c, s = c * scale, s * scale
            self._c, self._s = c, s

    def x(self):
        return self._x

    def y(self):
        return self._y

    def theta(self):
        return math.atan2(self._s, self._c)

    def compose(self, o):
        return Pose2(self._x + self._c * o._x - self._s * o._y, self._y + self._s * o._x + self._c * o._y,
                     _cs=(self._c * o._c - self._s * o._s, self._s * o._c + self._c * o._s))

    def inverse(self):
        return Pose2(-(self._c * self._x + self._s * self._y), -(-self._s * self._x + self._c * self._y),
                     _cs=(self._c, -self._s))

    def between(self, o):
        return self.inverse().compose(o)

    def matrix(self):
        return np.array([[self._c, -self._s, self._x], [self._s, self._c, self._y], [0.0, 0.0, 1.0]])


