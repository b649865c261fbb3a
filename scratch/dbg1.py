import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import distantspeech_amd as ds
from distantspeech_amd import _lib as L
from _cases import *
from oracle import ds_oracle as O
omic = oracle_mic(4, 512, 0.032)
x = np.stack([O.synth_utterance(50 + b, 256 * 60, omic) for b in range(3)])
a = steering(4, 512, 0.032)
def run(cuts):
    e = ds.BatchEngine(L.ALGO_ADAPTIVE, 4, 512, batch=3); e.set_steering(a)
    ys=[e.process(x[:, :, c0*256:c1*256], 1) for c0,c1 in zip(cuts[:-1],cuts[1:])]
    return np.concatenate(ys,axis=1), e.export_state()
y1,s1=run([0,60]); y1b,s1b=run([0,60])
print('determinism', np.array_equal(y1,y1b), np.array_equal(s1,s1b))
for cuts in ([0,25,60],[0,1,60],[0,59,60],[0,30,60],[0,2,4,60], list(range(61))):
    y,s=run(cuts)
    d=np.abs(y-y1); fr=np.unique(np.nonzero(d)[1]//256)
    print(cuts[:4],'equal',np.array_equal(y,y1),'state equal',np.array_equal(s,s1),'maxdiff',d.max(),'frames differing',fr[:10], len(fr))
