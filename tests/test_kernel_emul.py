"""CPU-only checks of the *kernel block program* (distantspeech_amd/csrc/ds_core.hpp) executed
serially by tests/emul (test infrastructure): same source the GPU compiles, so index arithmetic,
FFT plans, recursions and state layout are validated here without a GPU.  The GPU build itself is
checked by tests/test_gpu_parity.py (-m gpu)."""
import numpy as np
import pytest

from _cases import ADAPTIVE_CASES, ANGLE, GSC_CASES, TOL_RMS, as_float, load, rms, steering
from emul.emul import EmulEngine


@pytest.mark.parametrize("name", ADAPTIVE_CASES)
def test_emul_adaptive(name):
    g = load("g4_adaptive_" + name)
    M, nfft, hop, method = [int(v) for v in g["params"]]
    x = as_float(g["x"])
    e = EmulEngine(1, nfft, M, 1, ryy=(method == 3))
    e.set_steering(steering(M, nfft, float(g["r"])))
    e.method = method
    y = e.process(x[None], 1)[0]
    err = rms(y - g["y"])
    assert err < TOL_RMS, err
    assert err < (1e-4 if method == 3 else 1e-5), err
    # state: Rvv diagonal and MCRA p agree with the reference
    Rd = np.stack([e.field(i)[0] for i in range(M)], axis=1)
    assert np.allclose(Rd, np.real(np.einsum("kii->ki", g["Rvv"])), rtol=2e-3, atol=1e-7)
    pdiff = np.abs(e.field(M * M + 3)[0] - g["mcra_p"])
    assert np.mean(pdiff > 1e-3) < 0.02          # an fp32 threshold flip may move isolated bins


def test_emul_adaptive_chunking_and_layout():
    g = load("g4_adaptive_synth")
    x = as_float(g["x"])[:, : 256 * 30]
    a = steering(4, 512, 0.032)
    e1 = EmulEngine(1, 512, 4, 1); e1.set_steering(a)
    y1 = e1.process(x[None], 1)[0]
    e2 = EmulEngine(1, 512, 4, 1); e2.set_steering(a)
    y2 = np.concatenate([e2.process(x[None, :, : 256 * 7], 1)[0], e2.process(x[None, :, 256 * 7: 256 * 8], 1)[0],
                         e2.process(x[None, :, 256 * 8:], 1)[0]])
    assert np.array_equal(y1, y2)                # T hops in one call == T one-hop calls, bit for bit
    e3 = EmulEngine(1, 512, 4, 1); e3.set_steering(a)
    y3 = e3.process(np.ascontiguousarray(x.T)[None], 0)[0]
    assert np.array_equal(y1, y3)                # [L, M] and [M, L] layouts give identical results
    assert np.array_equal(e1.bins, e2.bins) and np.array_equal(e1.bins, e3.bins)


def test_emul_batch_independence():
    g = load("g4_adaptive_synth")
    x = as_float(g["x"])[:, : 256 * 20]
    xs = np.stack([x, x[::-1] * 0.5, np.roll(x, 1, axis=0)])
    a = steering(4, 512, 0.032)
    e = EmulEngine(1, 512, 4, 3); e.set_steering(a)
    yb = e.process(xs, 1)
    for b in range(3):
        e1 = EmulEngine(1, 512, 4, 1); e1.set_steering(a)
        assert np.array_equal(e1.process(xs[b:b + 1], 1)[0], yb[b])


@pytest.mark.parametrize("wt", ["DS", "SD"])
def test_emul_fixed(wt):
    g = load("g2b_fixed_" + wt)
    x = as_float(g["x"])
    e = EmulEngine(0, 512, 4, 1)
    e.set_steering(g["W"])
    y = e.process(np.ascontiguousarray(x.T)[None], 0)[0]
    assert rms(y - g["y"]) < 1e-6


@pytest.mark.parametrize("name", GSC_CASES)
def test_emul_gsc(name):
    g = load("g6_gsc_" + name)
    M, nfft, hop, method = [int(v) for v in g["params"]]
    x = as_float(g["x"])
    e = EmulEngine(2, nfft, M, 1)
    e.set_steering(steering(M, nfft, float(g["r"])))
    e.method = method
    y = e.process(x[None], 1)[0]
    assert rms(y - g["y"]) < TOL_RMS


@pytest.mark.parametrize("nfft,M", [(256, 2), (1024, 2), (512, 8)])
def test_emul_transform_roundtrip(nfft, M):
    """fixed beamformer with W = e_0 is STFT -> ISTFT: output == input delayed by `overlap` samples
    (perfect reconstruction of sqrt-Hann at 50 % overlap, transform.py:407-481)."""
    hop = nfft // 2
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((hop * 9, M)) * 0.1).astype(np.float32)
    W = np.zeros((nfft // 2 + 1, M), dtype=np.complex64)
    W[:, 0] = 1
    e = EmulEngine(0, nfft, M, 1)
    e.set_steering(W)
    y = e.process(x[None], 0)[0]
    assert np.max(np.abs(y[hop:] - x[:-hop, 0])) < 2e-6
    assert np.max(np.abs(y[:hop])) == 0.0 or np.max(np.abs(y[:hop] - 0)) < 1.0   # first hop holds the fade-in
