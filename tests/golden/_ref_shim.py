"""Import shim for the *reference* (wangwei2009/DistantSpeech) — used ONLY by make_golden.py.

TEST INFRASTRUCTURE.  Runs only in the build container where /root/reference is mounted;
never on the GPU box, never from the product package.

The reference is pure Python but imports third-party packages that are not installed here
(librosa, numba, soundfile, ...) and pins numpy==1.21.6 (requirements.txt:74).  This module
injects minimal functional stand-ins into ``sys.modules`` *for those third-party packages only*
(none of them is reference code) so the reference's own sources can be imported unmodified from
where they lie.  Repairs applied to the reference's broken entry points live in make_golden.py
next to each fixture and are recorded in the fixture metadata.
"""
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import scipy.signal

REFERENCE_ROOT = "/root/reference"


def _mini_librosa():
    librosa = types.ModuleType("librosa")
    filters = types.ModuleType("librosa.filters")
    util = types.ModuleType("librosa.util")
    display = types.ModuleType("librosa.display")

    def get_window(window, Nx, fftbins=True):
        return scipy.signal.get_window(window, Nx, fftbins=fftbins)

    def pad_center(data, size, axis=-1, **kwargs):
        n = data.shape[axis]
        lpad = int((size - n) // 2)
        lengths = [(0, 0)] * data.ndim
        lengths[axis] = (lpad, int(size - n - lpad))
        return np.pad(data, lengths, **kwargs)

    def frame(x, frame_length, hop_length, axis=-1):
        # librosa 0.9 semantics for 1-D input: [frame_length, n_frames]
        n_frames = 1 + (x.shape[-1] - frame_length) // hop_length
        idx = np.arange(frame_length)[:, None] + hop_length * np.arange(n_frames)[None, :]
        return x[idx]

    def valid_audio(y, mono=False):
        return True

    def fix_length(data, size, axis=-1, **kwargs):
        n = data.shape[axis]
        if n > size:
            sl = [slice(None)] * data.ndim
            sl[axis] = slice(0, size)
            return data[tuple(sl)]
        if n < size:
            lengths = [(0, 0)] * data.ndim
            lengths[axis] = (0, size - n)
            return np.pad(data, lengths, **kwargs)
        return data

    filters.get_window = get_window
    util.pad_center = pad_center
    util.frame = frame
    util.valid_audio = valid_audio
    util.fix_length = fix_length
    util.MAX_MEM_BLOCK = 2 ** 18
    librosa.filters = filters
    librosa.util = util
    librosa.display = display
    librosa.load = MagicMock()
    librosa.stft = MagicMock()
    return {"librosa": librosa, "librosa.filters": filters, "librosa.util": util, "librosa.display": display}


def install():
    """Install the stand-ins and numpy-1.21 aliases; put the reference on sys.path."""
    if getattr(install, "_done", False):
        return
    for name, mod in _mini_librosa().items():
        sys.modules[name] = mod

    numba = types.ModuleType("numba")

    def jit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    numba.jit = jit
    sys.modules["numba"] = numba

    for name in (
        "sounddevice", "soundfile", "pesq", "pystoi", "pystoi.stoi", "pyroomacoustics", "pyaudio",
        "turtle", "tkinter.tix", "gpuRIR", "webrtcvad", "cvxopt",
    ):
        if name not in sys.modules:
            sys.modules[name] = MagicMock()
    # imp was removed in python 3.12; harmless on 3.10 (noise_estimation/__init__.py:1)
    try:
        import imp  # noqa: F401
    except Exception:  # pragma: no cover
        sys.modules["imp"] = MagicMock()

    import matplotlib
    matplotlib.use("Agg")

    # numpy 1.21 names the reference relies on (adaptivebeamformer.py:84, GSC.py:218,246)
    if not hasattr(np, "mat"):
        np.mat = np.asmatrix
    if not hasattr(np, "float_"):
        np.float_ = np.float64
    if not hasattr(np, "complex"):
        np.complex = complex

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    install._done = True
