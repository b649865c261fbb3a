"""On-disk formats either side of the hot path: the reference's WAV / PCM helpers,
beamformer/utils.py:83-196.  Pure I/O — no signal processing happens here."""
import os

import numpy as np
from scipy.io import wavfile


def find_files(filepath, fileType: str):
    """all files of a type in a directory, in os.listdir order (utils.py:83-95)."""
    return [os.path.join(filepath, n) for n in os.listdir(filepath) if n.endswith(fileType)]


def load_wav(filepath):
    """load every .wav of a directory as one channel each -> ([M, L] float, sr)  (utils.py:98-124).
    The reference reads with librosa.load(sr=None): int16 PCM scaled by 1/32768, float32."""
    files = find_files(filepath, ".wav")
    chans, sr, min_len = [], None, None
    for name in files:
        sr, d = wavfile.read(name)
        if d.dtype == np.int16:
            d = d.astype(np.float32) / 32768.0
        elif d.dtype == np.int32:
            d = d.astype(np.float32) / 2147483648.0
        else:
            d = d.astype(np.float32)
        if d.ndim > 1:
            d = d.mean(axis=1)                     # librosa.load(mono=True)
        min_len = len(d) if min_len is None else min(min_len, len(d))
        chans.append(d)
    out = np.zeros([len(files), min_len])
    for i, d in enumerate(chans):
        out[i, :] = d[:min_len]
    return out, sr


def pcmread(filepath):
    """raw int16 PCM -> float in [-1, 1)  (utils.py:127-142)."""
    return np.memmap(filepath, dtype='h', mode='r') / 32768.0


def load_pcm(filepath):
    """every .pcm of a directory as one channel each -> [M, L]  (utils.py:145-163)."""
    files = find_files(filepath, ".pcm")
    data = [np.memmap(n, dtype='h', mode='r') / 32768.0 for n in files]
    out = np.zeros([len(files), len(data[0])])
    for i, d in enumerate(data):
        out[i, :] = d
    return out


def load_audio(filename: str) -> np.ndarray:
    """one WAV -> float32 scaled by 1/32767 (note: not 32768)  (utils.py:182-187)."""
    _, audio = wavfile.read(filename)
    if audio.dtype == np.int16:
        audio = audio.astype(np.float32) / float(np.iinfo(audio.dtype).max)
    return audio


def save_audio(filename: str, audio: np.ndarray, fs=16000):
    """float audio -> int16 WAV (x 32767, truncated)  (utils.py:190-196)."""
    if not filename.endswith(".wav"):
        filename = filename + ".wav"
    wavfile.write(filename, fs, (np.asarray(audio) * np.iinfo(np.int16).max).astype(np.int16))
