"""The plain-C restatement (oracle/c) against the reference's golden vectors G4 and against the NumPy oracle."""
import time

import numpy as np
import pytest

from _cases import ANGLE, as_float, load, rms, steering
from oracle.c_oracle import COracleMVDR


@pytest.mark.parametrize("name", ["rec1", "synth", "synth_m6", "synth_m8_1024"])
def test_c_oracle_vs_reference_golden(name):
    g = load("g4_adaptive_" + name)
    M, nfft, hop, method = [int(v) for v in g["params"]]
    assert method == 2
    x = as_float(g["x"])
    co = COracleMVDR(steering(M, nfft, float(g["r"])), nfft, hop)
    y = co.process(x)
    assert rms(y - g["y"]) < 1e-6 * max(rms(g["y"]), 1e-3)
    assert np.allclose(co.p, g["mcra_p"], rtol=1e-9, atol=1e-12)
