"""Array geometry — host-side set-up that runs once per look direction.

Mirrors the reference's beamformer/MicArray.py (MicArray :20-75, compute_tau :98-146 / :149-187,
steering_vector :77-96) and beamformer/gen_noise_msc.py:7-28.  Pure NumPy on the host: geometry
is not on the per-frame path."""
import numpy as np


def sph2cart(azimuth, elevation, r):
    x = r * np.cos(elevation) * np.cos(azimuth)
    y = r * np.cos(elevation) * np.sin(azimuth)
    z = r * np.sin(elevation)
    return x, y, z


class MicArray(object):
    """Same constructor arguments and attributes as the reference MicArray (room simulation excluded)."""

    def __init__(self, arrayType='circular', r=0.032, c=343, M=4, n_fft=256, mic_loc=None):
        self.arrayType = arrayType
        self.array_type = arrayType
        self.c = c
        self.r = r
        self.fs = 16000
        self.M = M
        self.n_fft = n_fft
        self.half_bin = round(self.n_fft / 2 + 1)
        self.freq_bin = np.linspace(0, self.half_bin - 1, self.half_bin)
        self.gamma = np.arange(0, 360, int(360 / self.M)) * np.pi / 180
        self.tau = np.zeros((self.M, 1))
        self.omega = 2 * np.pi * self.freq_bin * self.fs / self.n_fft
        self.mic_loc = np.zeros((M, 3))
        if arrayType == 'circular':
            az = np.arange(0, 360, int(360 / self.M)) * np.pi / 180
            for m in range(self.M):
                self.mic_loc[m, :] = sph2cart(az[m], 0, self.r)
        elif arrayType == 'linear':
            self.mic_loc[:, 0] = -(np.arange(self.M) - (self.M - 1) / 2) * self.r
        else:
            mic_loc = np.asarray(mic_loc, dtype=float)
            assert self.mic_loc.shape == mic_loc.shape, 'user defined mic location should be 2-D array with shape M X 3'
            self.mic_loc = mic_loc

    def compute_tau(self, incident_angle, normalize=False):
        """delays [M, 1] for an impinging direction given in radians (az, el)."""
        self.tau = compute_tau(self, np.asarray(incident_angle))
        if normalize:
            self.tau = self.tau - self.tau[0, 0]
        return self.tau

    def steering_vector(self, look_direction=0):
        """[half_bin, M] delay-only steering vector for an azimuth in degrees."""
        tau = self.compute_tau(np.array([look_direction, 0]) * np.pi / 180)
        return np.exp(-1j * self.omega[:, None] * tau[None, :, 0])


def compute_tau(mic_array, incident_angle):
    """Far-field delays [M, 1]: the projection of every microphone position on the unit vector of the
    impinging direction, over the speed of sound (reference MicArray.py:149-187, all microphones at once)."""
    ang = np.atleast_1d(np.asarray(incident_angle, dtype=float))
    direction = -np.array(sph2cart(ang[0], ang[1] if ang.size > 1 else 0.0, 1.0))      # unit vector towards the array
    pos = -np.asarray(mic_array.mic_loc, dtype=float)                                    # [M, 3]
    dist = np.linalg.norm(pos, axis=1)
    cosine = (pos * direction).sum(axis=1) / (np.linalg.norm(direction) * dist + 1e-12)
    return (-dist * cosine / mic_array.c)[:, None]


def gen_noise_msc(mic, nfft=256, Fvv_max=0.9998):
    """Diffuse-field coherence [half_bin, M, M]: sin(x) / x with x = 2 pi f d_ij / c off the diagonal,
    Fvv_max on it (reference gen_noise_msc.py:7-28)."""
    half_bin = round(nfft / 2 + 1)
    f = np.linspace(0, mic.fs / 2, half_bin)
    f[0] = 1e-6
    loc = np.asarray(mic.mic_loc, dtype=float)
    spacing = np.sqrt(((loc[:, None, :] - loc[None, :, :]) ** 2).sum(axis=2))            # [M, M]
    eye = np.eye(mic.M, dtype=bool)
    x = 2 * np.pi * f[:, None, None] * np.where(eye, 1.0, spacing)[None] / mic.c         # diagonal argument is a dummy
    return np.where(eye[None], Fvv_max, np.sin(x) / x)
