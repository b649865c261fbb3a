"""Host-side mirror of the reference's set-up code (geometry, fixed weights) against golden vectors."""
import numpy as np
import pytest

from _cases import load
from distantspeech_amd.mic_array import MicArray, compute_tau, gen_noise_msc
from distantspeech_amd.beamformer import beamformer


@pytest.mark.parametrize("name,atype", [("g2_weights_circular_M4_512", "circular"), ("g2_weights_linear_M6_512", "linear"),
                                        ("g2_weights_circular_M8_1024", "circular")])
def test_weights_match_reference(name, atype):
    g = load(name)
    M, nfft, az, el = g["params"]
    M, nfft = int(M), int(nfft)
    mic = MicArray(arrayType=atype, r=float(g["r"]), M=M, n_fft=nfft)
    assert np.allclose(mic.mic_loc, g["mic_loc"], atol=1e-15)
    assert np.allclose(mic.gamma, g["gamma"])
    assert np.allclose(gen_noise_msc(mic, nfft), g["Fvv"], atol=1e-12)
    bf = beamformer(mic, frame_len=nfft, hop=nfft // 2, nfft=nfft)
    assert np.allclose(bf.compute_steering_vector_from_doa((az, el)), g["a0"], atol=1e-12)
    assert np.allclose(bf.compute_weights([az, el], "DS"), g["Wds"], atol=1e-12)
    assert np.allclose(bf.compute_weights([az, el], "SD"), g["Wsd"], rtol=1e-9, atol=1e-9)


def test_compute_tau_identities():
    """the reference's own unit test identities (tests/unittests/test_micarray.py:5-32)."""
    mic = MicArray(arrayType="linear", M=4, r=0.032)
    tau = mic.compute_tau(np.array([0, 0]) / 180 * np.pi)
    assert abs((tau[-1, 0] - tau[0, 0]) * mic.c - (mic.M - 1) * mic.r) < 1e-6
    tau90 = compute_tau(mic, np.array([90, 0]) / 180 * np.pi)
    assert np.max(np.abs(tau90)) < 1e-12
    circ = MicArray(arrayType="circular", M=4, r=0.032)
    t = compute_tau(circ, np.array([0, 0]))
    assert abs(t[0, 0] * circ.c + circ.r) < 1e-9 and abs(t[2, 0] * circ.c - circ.r) < 1e-9
