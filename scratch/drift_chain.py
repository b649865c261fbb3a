"""fp32 drift of the two chain handles against the fp64 oracle over a long stream (error per 100-frame segment)."""
import sys, time
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import distantspeech_amd as ds
from oracle import ds_oracle as O
from _cases import ANGLE, oracle_mic, rms
def seg_err(y, ref, n):
    return " ".join("%.1e" % (rms(y[i:i + n] - ref[i:i + n]) / max(rms(ref[i:i + n]), 1e-9)) for i in range(0, len(ref), n))
# cfg4 chain
for M, nfft, T in ((4, 512, 1500), (8, 1024, 600)):
    hop = nfft // 2
    omic = oracle_mic(M, nfft)
    x = O.synth_utterance(21, T * hop, omic)
    t0 = time.time(); ref = O.OracleWpeMvdrPostfilter(omic, nfft=nfft, hop=hop).process(x, ANGLE); t1 = time.time()
    mic = ds.MicArray(arrayType="circular", r=omic.r, M=M, n_fft=nfft)
    obj = ds.WpeMvdrPostfilter(mic, frameLen=nfft, hop=hop)
    y = np.concatenate([obj.process(x[:, a:a + 50 * hop], ANGLE)["data"] for a in range(0, T * hop, 50 * hop)])
    print("cfg4 M=%d nfft=%d %d frames (oracle %.0fs): total rel %.2e | per 100 frames: %s" % (M, nfft, T, t1 - t0, rms(y - ref) / rms(ref), seg_err(y, ref, 100 * hop)), flush=True)
# cfg5 chain (RLS blocking filters)
M, FL, T = 6, 256, 1200
omic = oracle_mic(M, 512)
x = O.synth_utterance(22, T * FL, omic) * 0.1
og = O.OracleSubbandGSC(omic, frameLen=FL, rls_bm=True)
with np.errstate(all="ignore"):
    ref = og.process(x)[0]
mic = ds.MicArray(arrayType="circular", r=omic.r, M=M, n_fft=512)
sg = ds.SubbandGSC(mic, frameLen=FL, bm_filter="rls")
y = np.concatenate([sg.process(x[:, a:a + 100 * FL])[0] for a in range(0, T * FL, 100 * FL)])
print("cfg5 M=6 RLS %d blocks: total rel %.2e | per 100 blocks: %s" % (T, rms(y - ref) / rms(ref), seg_err(y, ref, 100 * FL)), flush=True)
