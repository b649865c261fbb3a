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
    incident_angle = np.asarray(incident_angle, dtype=float)
    az = incident_angle[0]
    el = incident_angle[1] if incident_angle.ndim > 0 and incident_angle.size > 1 else 0
    x0, y0, z0 = sph2cart(az, el, 1)
    p0 = -1 * np.array([x0, y0, z0])
    tau = np.zeros((mic_array.M, 1))
    for m in range(mic_array.M):
        mic_loc_m = -1 * mic_array.mic_loc[m, :]
        nrm = np.linalg.norm(mic_loc_m)
        cos_theta = np.sum(mic_loc_m * p0) / (np.linalg.norm(p0) * nrm + 1e-12)
        tau[m] = -1 * nrm * cos_theta / mic_array.c
    return tau


def gen_noise_msc(mic, nfft=256, Fvv_max=0.9998):
    """Diffuse-field coherence matrix [half_bin, M, M] (sinc model)."""
    M, c, fs = mic.M, mic.c, mic.fs
    half_bin = round(nfft / 2 + 1)
    Fvv = np.zeros((half_bin, M, M))
    f = np.linspace(0, fs / 2, half_bin)
    f[0] = 1e-6
    for i in range(M):
        for j in range(M):
            if i == j:
                Fvv[:, i, j] = Fvv_max
            else:
                dij = np.sqrt(np.sum((mic.mic_loc[i, :] - mic.mic_loc[j, :]) ** 2))
                Fvv[:, i, j] = np.sin(2 * np.pi * f * dij / c) / (2 * np.pi * f * dij / c)
    return Fvv
