"""Single-process stand-in for mpi4py (absent here; test infrastructure, build container only).
The reference's trainer uses COMM_WORLD Allreduce / Bcast / Gather (utils/mpi_tools.py:117-187);
with one rank these are copies."""
import numpy as np


class _Op:
    def __init__(self, name):
        self.name = name


class _Comm:
    def Get_rank(self):
        return 0

    def Get_size(self):
        return 1

    def Allreduce(self, sendbuf, recvbuf, op=None):
        np.copyto(np.asarray(recvbuf), np.asarray(sendbuf))

    def Bcast(self, x, root=0):
        return None

    def Gather(self, sendbuf, recvbuf, root=0):
        np.copyto(np.asarray(recvbuf).reshape(np.asarray(sendbuf).shape), np.asarray(sendbuf))


class MPI:
    COMM_WORLD = _Comm()
    SUM, MIN, MAX = _Op("sum"), _Op("min"), _Op("max")
