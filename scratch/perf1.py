import sys, time, numpy as np, torch
sys.path.insert(0,'.')
from distantspeech_amd import BatchEngine, _lib as L
import bench
M,NFFT,HOP=4,512,256
dev=torch.device('cuda',0)
def setup(B, hops, algo=L.ALGO_ADAPTIVE):
    Ltot=hops*HOP
    x=(torch.randn((B,M,Ltot),device=dev)*0.05)
    y=torch.empty((B,Ltot),device=dev)
    eng=BatchEngine(algo,M,NFFT,HOP,batch=B,device=0)
    from distantspeech_amd.mic_array import MicArray
    mic=MicArray(M=M,n_fft=NFFT)
    tao=-1*mic.r*np.cos(0)*np.cos(bench.ANGLE[0]-mic.gamma)/mic.c
    omega=2*np.pi*np.arange(257)*16000/512
    eng.set_steering(np.exp(-1j*omega[:,None]*tao[None,:])); eng.set_method(2)
    torch.cuda.synchronize()
    return x,y,eng,Ltot
def timeit(eng, fn, reps=3):
    best=1e9
    for _ in range(reps):
        eng.synchronize(); eng.timing_begin(); fn(); d=eng.timing_end()
        best=min(best,d)
    return best
for algo,name in ((L.ALGO_ADAPTIVE,'adaptive'),(L.ALGO_GSC,'gsc'),(L.ALGO_FIXED,'fixed')):
  for B in (1024, 4096):
    K=400
    x,y,eng,Ltot=setup(B,K,algo)
    xp,yp=x.data_ptr(),y.data_ptr()
    for T in (1, 4, 16, 400):
        n=K//T
        for graph in ((0,1) if T==1 else (0,)):
            if graph: eng.process_device_seq(xp,1,M*Ltot,Ltot,T*HOP,T*HOP,n,yp,Ltot,T*HOP,graph=2)
            def cl():
                eng.process_device_seq(xp,1,M*Ltot,Ltot,T*HOP,T*HOP,n,yp,Ltot,T*HOP,graph=graph)
            d=timeit(eng,cl); print('%s B %d T=%d graph=%d: %.2f us/call  %.1f Mframes/s'%(name,B,T,graph,d/n*1e3,B*n*T/d/1e3), flush=True)
    del x,y,eng
