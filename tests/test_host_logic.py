"""Host-side mirror of the reference's set-up code (geometry, fixed weights) against golden vectors."""
import numpy as np
import pytest

from _cases import load
from oracle import ds_oracle as O
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


def test_front_end_tables_match_the_oracle():
    """host-side set-up of the chains: fractional-delay FIR bank and the McCDR diffuse coherence table."""
    from distantspeech_amd.subband_gsc import fractional_delay_filter_bank
    from distantspeech_amd.ops import McSpp
    from distantspeech_amd.mic_array import MicArray, compute_tau
    for M, r in ((4, 0.032), (6, 0.05)):
        mic = MicArray(arrayType="circular", r=r, M=M, n_fft=512)
        omic = O.OracleMicArray(arrayType="circular", r=r, M=M, n_fft=512)
        ang = np.array([197, 0]) / 180 * np.pi
        tau = compute_tau(mic, ang)
        assert np.allclose(tau, O.compute_tau(omic, ang), atol=1e-15)
        d = np.array(-(tau - np.max(tau)))[:, 0] * 16000
        assert np.array_equal(fractional_delay_filter_bank(d), O.fractional_delay_filter_bank(d))
        Fn = McSpp.diffuse_coherence(M, 512)
        assert Fn.shape == (257,) and np.allclose(Fn, O.gen_noise_msc(O.OracleMicArray(arrayType="circular", r=0.032, M=M), 512)[:, 1, 2])


def test_unsupported_configurations_fail_before_touching_the_device():
    """argument errors are raised by the mirrors themselves (no GPU needed to see them)."""
    import distantspeech_amd as ds
    with pytest.raises(NotImplementedError):
        ds.AdaptiveBlockingMatrixFilter(filter_len=64, two_path=True)           # two_path belongs to the plain FastFreqLms
    with pytest.raises(NotImplementedError):
        ds.FastFreqLms(filter_len=100)
    with pytest.raises(NotImplementedError):
        ds.FastFreqLms(filter_len=64, hop_len=32)
    with pytest.raises(NotImplementedError):
        ds.Transform(channel=2, n_fft=512, hop_length=256, window=np.ones(300))   # a custom window must have n_fft samples


def test_wav_helpers_roundtrip(tmp_path):
    """save_audio / load_audio / load_wav ordering and scaling (beamformer/utils.py) without a GPU."""
    from distantspeech_amd import utils
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(4000) * 0.1).astype(np.float32)
    path = str(tmp_path / "a.wav")
    utils.save_audio(path, x, 16000)
    y = utils.load_audio(path)
    assert y.shape == x.shape and np.max(np.abs(y - x)) < 2.0 / 32767
