"""Shared helpers for the parity tests (golden fixtures -> inputs, steering vectors)."""
import os

import numpy as np

from oracle import ds_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ANGLE = np.array([197, 0]) / 180 * np.pi

# north-star tolerance: output within 1e-4 RMS of the NumPy reference (BASELINE.json);
# the fp32 kernels are expected (and asserted) to be ~50x better than that on these inputs.
TOL_RMS = 1e-4


def rms(a):
    a = np.asarray(a)
    return float(np.sqrt(np.mean(np.abs(a) ** 2)))


def measured(test, **values):
    """Record the measured parity numbers of a GPU test (what the assertions bound): one JSON line per call appended to
    $DS_PARITY_LOG (default gpurun_out/parity_measured.jsonl); the per-round copy is committed under profiles/."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.environ.get("DS_PARITY_LOG") or os.path.join(root, "gpurun_out", "parity_measured.jsonl")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "a") as fh:
            fh.write(json.dumps(dict(test=test, **{k: (float(v) if np.ndim(v) == 0 else [float(u) for u in v]) for k, v in values.items()})) + "\n")
    except OSError:
        pass


def relmax(a, ref):
    """max |a - ref| over max |ref|"""
    return float(np.max(np.abs(np.asarray(a) - np.asarray(ref))) / (np.max(np.abs(ref)) + 1e-300))


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def as_float(x):
    return x.astype(np.float32) / 32768.0 if x.dtype == np.int16 else x.astype(np.float32)


def oracle_mic(M, nfft, r=None, atype="circular"):
    return O.OracleMicArray(arrayType=atype, r=(0.032 if M == 4 else 0.05) if r is None else r, M=M, n_fft=nfft)


def steering(M, nfft, r, angle=ANGLE):
    mic = oracle_mic(M, nfft, r)
    tao = O.circular_tao(mic.r, mic.c, mic.gamma, angle)
    omega = 2 * np.pi * np.arange(nfft // 2 + 1) * 16000 / nfft
    return np.exp(-1j * omega[:, None] * tao[None, :])


ADAPTIVE_CASES = ["rec1", "synth", "synth_ds", "synth_src", "synth_tfgsc", "synth_m6", "synth_m8_1024", "synth_m3", "synth_m5"]
GSC_CASES = ["rec1", "synth_m6", "synth_m4", "synth_m0", "synth_m3", "synth_m5"]


class DeviceBuffers:
    """Raw device allocations for the tests that drive the device-pointer entry points (hipMalloc / hipMemcpy through ctypes on the HIP
    runtime libdsenh.so is linked against — no second runtime in the process)."""

    def __init__(self):
        import ctypes
        self.ct = ctypes
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.ptrs = []

    def _ok(self, rc, what):
        assert rc == 0, "%s failed with hipError %d" % (what, rc)

    def upload(self, a):
        a = np.ascontiguousarray(a)
        p = self.ct.c_void_p()
        self._ok(self.hip.hipMalloc(self.ct.byref(p), self.ct.c_size_t(a.nbytes)), "hipMalloc")
        self._ok(self.hip.hipMemcpy(p, a.ctypes.data_as(self.ct.c_void_p), self.ct.c_size_t(a.nbytes), 1), "hipMemcpy H2D")
        self.ptrs.append(p)
        return p.value

    def zeros(self, nbytes):
        p = self.ct.c_void_p()
        self._ok(self.hip.hipMalloc(self.ct.byref(p), self.ct.c_size_t(nbytes)), "hipMalloc")
        self._ok(self.hip.hipMemset(p, 0, self.ct.c_size_t(nbytes)), "hipMemset")
        self.ptrs.append(p)
        return p.value

    def download(self, ptr, shape, dtype=np.float32):
        out = np.empty(shape, dtype=dtype)
        self._ok(self.hip.hipDeviceSynchronize(), "hipDeviceSynchronize")
        self._ok(self.hip.hipMemcpy(out.ctypes.data_as(self.ct.c_void_p), self.ct.c_void_p(ptr), self.ct.c_size_t(out.nbytes), 2), "hipMemcpy D2H")
        return out

    def free(self):
        for p in self.ptrs:
            self.hip.hipFree(p)
        self.ptrs = []
