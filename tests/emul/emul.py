"""ctypes wrapper of the serial CPU execution of the kernel block program (TEST INFRASTRUCTURE).

Builds tests/emul/libds_emul.so from tests/emul/ds_emul.cpp + distantspeech_amd/csrc/ds_core.hpp with
g++ on first use.  Used only by tests/test_kernel_emul.py to check the kernel's arithmetic against
the oracle in the CPU-only container; never imported by the product package."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SO = os.path.join(HERE, "libds_emul.so")
SRCS = [os.path.join(HERE, "ds_emul.cpp"), os.path.join(ROOT, "distantspeech_amd", "csrc", "ds_core.hpp"),
        os.path.join(ROOT, "distantspeech_amd", "csrc", "ds_tables.hpp")]


def build(force=False):
    if not force and os.path.exists(SO) and all(os.path.getmtime(SO) >= os.path.getmtime(s) for s in SRCS):
        return SO
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-mfma", SRCS[0], "-o", SO])
    return SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.emul_layout.restype = ctypes.c_int
        _lib.emul_run.restype = ctypes.c_int
    return _lib


class EmulEngine:
    """Holds the per-utterance state arrays the GPU keeps in HBM and runs the block program on them."""

    def __init__(self, algo, nfft, M, batch=1, ryy=False, mcra_L=15):
        self.algo, self.nfft, self.M, self.batch, self.ryy = algo, nfft, M, batch, int(ryy)
        self.hop, self.K = nfft // 2, nfft // 2 + 1
        kp = ctypes.c_int(0)
        self.NP = lib().emul_layout(algo, nfft, M, self.ryy, ctypes.byref(kp))
        self.KP = kp.value
        self.bins = np.zeros((batch, max(self.NP, 1), self.KP, 4), dtype=np.float32)
        self.tail_in = np.zeros((batch, M, self.hop), dtype=np.float32)
        self.tail_out = np.zeros((batch, self.hop), dtype=np.float32)
        self.counters = np.zeros((batch, 4), dtype=np.int32)
        self.counters[:, 1] = 1
        self.steer = None
        self.method, self.mcra_L = 2, mcra_L
        self.alpha_y, self.alpha_v, self.diag, self.gate, self.mu = 0.8, 0.9998, 1e-6, 0.4, 0.01

    def set_steering(self, a):
        a = np.ascontiguousarray(a, dtype=np.complex64)
        self.steer_per_utt = int(a.ndim == 3)
        self.steer = a

    def process(self, x, layout):
        """x [B, L, M] (layout 0) or [B, M, L] (layout 1) float32 -> y [B, L]"""
        x = np.ascontiguousarray(x, dtype=np.float32)
        L = x.shape[1] if layout == 0 else x.shape[2]
        y = np.zeros((self.batch, L), dtype=np.float32)
        f = ctypes.c_float
        rc = lib().emul_run(self.algo, self.nfft, self.M, self.ryy, self.batch, x.ctypes.data_as(ctypes.c_void_p), layout, L,
                            y.ctypes.data_as(ctypes.c_void_p), self.bins.ctypes.data_as(ctypes.c_void_p),
                            self.tail_in.ctypes.data_as(ctypes.c_void_p), self.tail_out.ctypes.data_as(ctypes.c_void_p),
                            self.counters.ctypes.data_as(ctypes.c_void_p), self.steer.ctypes.data_as(ctypes.c_void_p),
                            self.steer_per_utt, self.method, self.mcra_L, f(self.alpha_y), f(self.alpha_v), f(self.diag),
                            f(self.gate), f(self.mu))
        assert rc == 0, rc
        return y

    def field(self, f):
        """float index f of every bin -> [B, K]"""
        return self.bins[:, f // 4, : self.K, f % 4]
