"""BASELINE cfg5 batch (2048 utterances), 60 single-block device calls enqueued back to back + a few multi-block ones: the pipelined chain
against the serial chain (DS_CHAIN_SERIAL_FRONT=1), samples and exported state bit for bit.  usage: python scratch/stress_pipeline.py [B]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
M, FL = 6, 256
res = []
for serial in ("1", "0", "0"):
    os.environ["DS_CHAIN_SERIAL_FRONT"] = serial
    import distantspeech_amd as ds
    from distantspeech_amd import _lib as L
    from _cases import DeviceBuffers
    mic = ds.MicArray(arrayType="circular", r=0.05, M=M, n_fft=2 * FL)
    rng = np.random.default_rng(5)
    nblk = 80
    x = (rng.standard_normal((B, M, nblk * FL)) * 0.05).astype(np.float32)
    dv = DeviceBuffers()
    xd, yd = dv.upload(x), dv.zeros(B * nblk * FL * 4)
    g = ds.SubbandGSC(mic, frameLen=FL, angle=[197, 0], batch=B, bm_filter="rls")
    e = g._eng
    Ltot = nblk * FL
    pos = 0
    for n_calls, T in ((60, 1), (2, 5), (10, 1)):
        e.process_device_seq(xd + 4 * pos * FL, L.LAYOUT_CHANNELS_SAMPLES, M * Ltot, Ltot, T * FL, T * FL, n_calls, yd + 4 * pos * FL, Ltot, T * FL, graph=0)
        pos += n_calls * T
    y = dv.download(yd, (B, Ltot))
    st = np.frombuffer(e.export_state(), dtype=np.float32).copy()
    res.append((y, st))
    dv.free(); e.close()
    print("serial=%s done, |y| max %.4f" % (serial, np.abs(y).max()), flush=True)
ok = all(np.array_equal(res[0][0], r[0]) and np.array_equal(res[0][1], r[1]) for r in res[1:])
print("pipelined == serial:", ok)
sys.exit(0 if ok else 1)
