"""Stand-in for `pybullet_utils.bullet_client` (test infrastructure, this container only).

On the SimplePhysics path the Bullet world is only a mirror of the Python state
(reference envs/physics.py:190-200) plus a pose/velocity round trip at reset
(envs/agents.py:434-453), so a dict of base pose/velocity per body is sufficient -- with ONE
piece of Bullet arithmetic restated: the base orientation is read back through btTransform's
3x3 basis (see getBasePositionAndOrientation), which canonicalises the quaternion's sign.
"""
import math as _math

import pybullet as _pb


def _bt_set_rotation(q):
    """btMatrix3x3::setRotation(const btQuaternion&) (LinearMath/btMatrix3x3.h), scalar path."""
    x, y, z, w = q
    d = x * x + y * y + z * z + w * w
    s = 2.0 / d
    xs, ys, zs = x * s, y * s, z * s
    wx, wy, wz = w * xs, w * ys, w * zs
    xx, xy, xz = x * xs, x * ys, x * zs
    yy, yz, zz = y * ys, y * zs, z * zs
    return ((1.0 - (yy + zz), xy - wz, xz + wy),
            (xy + wz, 1.0 - (xx + zz), yz - wx),
            (xz - wy, yz + wx, 1.0 - (xx + yy)))


def _bt_get_rotation(m):
    """btMatrix3x3::getRotation(btQuaternion&) (LinearMath/btMatrix3x3.h), scalar path."""
    trace = m[0][0] + m[1][1] + m[2][2]
    temp = [0.0, 0.0, 0.0, 0.0]
    if trace > 0.0:
        s = _math.sqrt(trace + 1.0)
        temp[3] = s * 0.5
        s = 0.5 / s
        temp[0] = (m[2][1] - m[1][2]) * s
        temp[1] = (m[0][2] - m[2][0]) * s
        temp[2] = (m[1][0] - m[0][1]) * s
    else:
        i = (2 if m[1][1] < m[2][2] else 1) if m[0][0] < m[1][1] else (2 if m[0][0] < m[2][2] else 0)
        j = (i + 1) % 3
        k = (i + 2) % 3
        s = _math.sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0)
        temp[i] = s * 0.5
        s = 0.5 / s
        temp[3] = (m[k][j] - m[j][k]) * s
        temp[j] = (m[j][i] + m[i][j]) * s
        temp[k] = (m[k][i] + m[i][k]) * s
    return tuple(temp)


class BulletClient:
    COV_ENABLE_GUI = _pb.COV_ENABLE_GUI
    COV_ENABLE_RENDERING = _pb.COV_ENABLE_RENDERING
    GEOM_SPHERE = _pb.GEOM_SPHERE

    def __init__(self, connection_mode=None):
        self._next = 0
        self._pose = {}
        self._vel = {}
        self._saved = {}

    def _new_body(self, pos=(0., 0., 0.), orn=(0., 0., 0., 1.)):
        bid = self._next
        self._next += 1
        self._pose[bid] = (tuple(float(v) for v in pos), tuple(float(v) for v in orn))
        self._vel[bid] = ((0., 0., 0.), (0., 0., 0.))
        return bid

    def loadURDF(self, fileName, basePosition=(0., 0., 0.), baseOrientation=(0., 0., 0., 1.), **kw):
        return self._new_body(basePosition, baseOrientation)

    def createMultiBody(self, basePosition=(0., 0., 0.), **kw):
        return self._new_body(basePosition)

    def createVisualShape(self, *a, **kw):
        return 0

    def resetBasePositionAndOrientation(self, bid, posObj, ornObj):
        self._pose[bid] = (tuple(float(v) for v in posObj), tuple(float(v) for v in ornObj))

    def getBasePositionAndOrientation(self, bid):
        """Bullet does NOT hand the quaternion back verbatim.  Published Bullet3 code path
        (pybullet is unpinned in the reference's setup.py:32; files of bullet3 master / 3.2x):
          * examples/pybullet/pybullet.c `pybullet_internalGetBasePositionAndOrientation` copies
            `actualStateQ[0..6]` of the CMD_REQUEST_ACTUAL_STATE status, nothing else;
          * examples/SharedMemory/PhysicsServerCommandProcessor.cpp
            `processRequestActualStateCommand` fills them for a btMultiBody base (loadURDF's
            default) with `btTransform tr; tr.setOrigin(mb->getBasePos());
            tr.setRotation(mb->getWorldToBaseRot().inverse());` ... `tr.getRotation()[i]`
            (the root inertial frame of cf21x_sys_eq.urdf:14-18 is the identity);
            `processInitPoseCommand` had stored `setWorldToBaseRot(q.inverse())` -- conjugation
            twice is exact;
          * src/LinearMath/btTransform.h `setRotation` / `getRotation` go through the 3x3 basis:
            btMatrix3x3::setRotation(q) then btMatrix3x3::getRotation(q) (btMatrix3x3.h), in
            double precision (pybullet is built with BT_USE_DOUBLE_PRECISION).
        So the quaternion that comes back is the matrix->quaternion extraction: w > 0 when the
        trace is positive, else the component of the largest diagonal element positive; the
        position is returned as stored."""
        pos, orn = self._pose[bid]
        return pos, _bt_get_rotation(_bt_set_rotation(orn))

    def resetBaseVelocity(self, bid, linearVelocity=None, angularVelocity=None):
        lin, ang = self._vel[bid]
        if linearVelocity is not None:
            lin = tuple(float(v) for v in linearVelocity)
        if angularVelocity is not None:
            ang = tuple(float(v) for v in angularVelocity)
        self._vel[bid] = (lin, ang)

    def getBaseVelocity(self, bid):
        return self._vel[bid]

    def getLinkStates(self, bid, linkIndices=(), **kw):
        """World position of the URDF links of the CrazyFlie model: the four motor links are fixed
        at (+-0.028, +-0.028, 0) in the base frame (reference envs/assets/cf21x_sys_eq.urdf:47,59,
        71,83), link 4 is the centre of mass.  Only element [i][0] (world position) is consumed by
        BasePhysics.calculate_ground_effect (reference envs/physics.py:45-50)."""
        offs = [(0.028, -0.028, 0.), (-0.028, -0.028, 0.), (-0.028, 0.028, 0.), (0.028, 0.028, 0.),
                (0., 0., 0.)]
        pos, orn = self._pose[bid]
        R = _pb.getMatrixFromQuaternion(orn)
        out = []
        for i in linkIndices:
            o = offs[i]
            w = tuple(pos[r] + R[3 * r + 0] * o[0] + R[3 * r + 1] * o[1] + R[3 * r + 2] * o[2]
                      for r in range(3))
            out.append((w, orn))
        return out

    def saveState(self):
        sid = len(self._saved)
        self._saved[sid] = (dict(self._pose), dict(self._vel))
        return sid

    def restoreState(self, sid):
        p, v = self._saved[sid]
        self._pose, self._vel = dict(p), dict(v)

    # pure math passthroughs
    getQuaternionFromEuler = staticmethod(_pb.getQuaternionFromEuler)
    getMatrixFromQuaternion = staticmethod(_pb.getMatrixFromQuaternion)
    getEulerFromQuaternion = staticmethod(_pb.getEulerFromQuaternion)

    def __getattr__(self, name):
        # everything else (debug visualiser, gravity, engine params, changeDynamics ...) is a no-op
        if name.startswith("__"):  # introspection (the reference's JSON logger probes __name__), not a Bullet call
            raise AttributeError(name)

        def _noop(*a, **kw):
            return 0
        return _noop
