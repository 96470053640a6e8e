"""Stand-in for `pybullet_data` (test infrastructure, this container only)."""


def getDataPath():
    return "/nonexistent"
