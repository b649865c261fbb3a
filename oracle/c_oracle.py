"""ctypes wrapper of oracle/c/libds_oracle_c.so — plain-C double-precision restatement of the reference's adaptive-MVDR
frame loop.  TEST INFRASTRUCTURE (see oracle/ds_oracle.py): checker and timed CPU baseline only."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "c", "libds_oracle_c.so")


def build():
    src = os.path.join(HERE, "c", "ds_oracle_mvdr.c")
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(HERE, "c")])
    return SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.dso_create.restype = ctypes.c_void_p
        _lib.dso_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        _lib.dso_destroy.argtypes = [ctypes.c_void_p]
        _lib.dso_process.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        _lib.dso_get_p.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    return _lib


class COracleMVDR:
    def __init__(self, steer, nfft=512, hop=256, mcra_L=15):
        steer = np.ascontiguousarray(steer, dtype=np.complex128)
        self.K, self.M = steer.shape
        self.hop = hop
        self._h = lib().dso_create(self.M, nfft, hop, steer.ctypes.data_as(ctypes.c_void_p), mcra_L)

    def process(self, x):
        """x [M, T*hop] float32 -> y [T*hop] float32"""
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.zeros(x.shape[1], dtype=np.float32)
        lib().dso_process(self._h, x.ctypes.data_as(ctypes.c_void_p), x.shape[1], y.ctypes.data_as(ctypes.c_void_p))
        return y

    @property
    def p(self):
        out = np.zeros(self.K)
        lib().dso_get_p(self._h, out.ctypes.data_as(ctypes.c_void_p))
        return out

    def __del__(self):
        try:
            lib().dso_destroy(self._h)
        except Exception:
            pass
