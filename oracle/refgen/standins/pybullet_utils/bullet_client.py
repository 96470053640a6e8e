"""Stand-in for `pybullet_utils.bullet_client` (test infrastructure, this container only).

On the SimplePhysics path the Bullet world is only a mirror of the Python state
(reference envs/physics.py:190-200) plus a pose/velocity round trip at reset
(envs/agents.py:434-453), so a dict of base pose/velocity per body is sufficient.
"""
import pybullet as _pb


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
        return self._pose[bid]

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
        def _noop(*a, **kw):
            return 0
        return _noop
