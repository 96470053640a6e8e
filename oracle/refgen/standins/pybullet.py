"""Stand-in for the third-party `pybullet` module (absent from this image, no network).

TEST INFRASTRUCTURE ONLY -- used in THIS container by oracle/refgen/gen_golden.py so that the
reference's `phoenix_drone_simulation.envs` package can be imported from /root/reference.
It never travels to the GPU box as part of the product path and nothing in the product imports it.

Only the three pure math functions the SimplePhysics path calls are real; they restate the
published Bullet3 algorithms (pybullet is listed unpinned in the reference's setup.py:32):
  * getQuaternionFromEuler  -- Bullet3 examples/pybullet/pybullet.c `pybullet_getQuaternionFromEuler`
                               (ZYX half-angle products, order [x,y,z,w], then normalised); the same
                               formula is restated in the reference at envs/utils.py:32-56.
  * getMatrixFromQuaternion -- b3Matrix3x3::setRotation (s = 2/|q|^2), row-major 9-tuple.
  * getEulerFromQuaternion  -- pybullet.c `pybullet_getEulerFromQuaternion` (gimbal guard at
                               |sarg| >= 0.99999).
Call sites in the reference: envs/physics.py:160,179; envs/agents.py:55,446,452; envs/hover.py:209,237.
"""
import math

GUI = 1
DIRECT = 2
LINK_FRAME = 1
WORLD_FRAME = 2
COV_ENABLE_GUI = 1
COV_ENABLE_RENDERING = 7
GEOM_SPHERE = 2
URDF_USE_INERTIA_FROM_FILE = 2
VELOCITY_CONTROL = 0


def getQuaternionFromEuler(rpy):
    phi, the, psi = rpy[0] * 0.5, rpy[1] * 0.5, rpy[2] * 0.5
    sphi, cphi = math.sin(phi), math.cos(phi)
    sthe, cthe = math.sin(the), math.cos(the)
    spsi, cpsi = math.sin(psi), math.cos(psi)
    x = sphi * cthe * cpsi - cphi * sthe * spsi
    y = cphi * sthe * cpsi + sphi * cthe * spsi
    z = cphi * cthe * spsi - sphi * sthe * cpsi
    w = cphi * cthe * cpsi + sphi * sthe * spsi
    n = math.sqrt(x * x + y * y + z * z + w * w)
    return (x / n, y / n, z / n, w / n)


def getMatrixFromQuaternion(q):
    x, y, z, w = q
    d = x * x + y * y + z * z + w * w
    s = 2.0 / d
    xs, ys, zs = x * s, y * s, z * s
    wx, wy, wz = w * xs, w * ys, w * zs
    xx, xy, xz = x * xs, x * ys, x * zs
    yy, yz, zz = y * ys, y * zs, z * zs
    return (1.0 - (yy + zz), xy - wz, xz + wy,
            xy + wz, 1.0 - (xx + zz), yz - wx,
            xz - wy, yz + wx, 1.0 - (xx + yy))


def getEulerFromQuaternion(q):
    x, y, z, w = q
    sqx, sqy, sqz, squ = x * x, y * y, z * z, w * w
    sarg = -2.0 * (x * z - w * y)
    if sarg <= -0.99999:
        return (0.0, -0.5 * math.pi, 2.0 * math.atan2(x, -y))
    if sarg >= 0.99999:
        return (0.0, 0.5 * math.pi, 2.0 * math.atan2(-x, y))
    return (math.atan2(2.0 * (y * z + w * x), squ - sqx - sqy + sqz),
            math.asin(sarg),
            math.atan2(2.0 * (x * y + w * z), squ + sqx - sqy - sqz))


def loadURDF(*args, **kwargs):
    return 99
