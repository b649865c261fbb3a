import sys, time, numpy as np, torch
sys.path.insert(0,'.')
from distantspeech_amd import BatchEngine, _lib as L
import bench
M,NFFT,HOP=4,512,256
dev=torch.device('cuda',0)
def setup(B, hops):
    Ltot=hops*HOP
    x=(torch.randn((B,M,Ltot),device=dev)*0.05)
    y=torch.empty((B,Ltot),device=dev)
    eng=BatchEngine(L.ALGO_ADAPTIVE,M,NFFT,HOP,batch=B,device=0)
    from distantspeech_amd.mic_array import MicArray
    mic=MicArray(M=M,n_fft=NFFT)
    tao=-1*mic.r*np.cos(0)*np.cos(bench.ANGLE[0]-mic.gamma)/mic.c
    omega=2*np.pi*np.arange(257)*16000/512
    eng.set_steering(np.exp(-1j*omega[:,None]*tao[None,:])); eng.set_method(2)
    return x,y,eng,Ltot
stream=torch.cuda.current_stream().cuda_stream
def timeit(fn, n_frames, reps=3):
    best=1e9
    for _ in range(reps):
        torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        t0=time.perf_counter(); e0.record(); fn(); e1.record(); torch.cuda.synchronize(); t=time.perf_counter()-t0
        best=min(best,t); d=e0.elapsed_time(e1)
    return best, d
for B in (1024, 4096):
    K=400
    x,y,eng,Ltot=setup(B,K)
    xp,yp=x.data_ptr(),y.data_ptr()
    def pyloop():
        for i in range(K):
            eng.process_device(xp+4*i*HOP, 1, M*Ltot, HOP, yp+4*i*HOP, Ltot, stream=stream, x_chan_stride=Ltot)
    t,d=timeit(pyloop,B*K); print('B',B,'python loop T=1: host %.1f us/step dev %.1f us/step  %.1f Mframes/s'%(t/K*1e6,d/K*1e3,B*K/t/1e6))
    def cloop():
        eng.process_device_seq(xp,1,M*Ltot,Ltot,HOP,HOP,K,yp,Ltot,HOP,stream=stream,graph=0)
    t,d=timeit(cloop,B*K); print('B',B,'C loop T=1: host %.1f us/step dev %.1f us/step  %.1f Mframes/s'%(t/K*1e6,d/K*1e3,B*K/t/1e6))
    eng.process_device_seq(xp,1,M*Ltot,Ltot,HOP,HOP,K,yp,Ltot,HOP,stream=stream,graph=2)
    def graph():
        eng.process_device_seq(xp,1,M*Ltot,Ltot,HOP,HOP,K,yp,Ltot,HOP,stream=stream,graph=1)
    t,d=timeit(graph,B*K); print('B',B,'graph T=1: host %.1f us/step dev %.1f us/step  %.1f Mframes/s'%(t/K*1e6,d/K*1e3,B*K/t/1e6))
    for T in (4, 16, 400):
        n=K//T
        def cl():
            eng.process_device_seq(xp,1,M*Ltot,Ltot,T*HOP,T*HOP,n,yp,Ltot,T*HOP,stream=stream,graph=0)
        t,d=timeit(cl,B*n*T); print('B',B,'T=%d C loop: %.1f us/call dev  %.1f Mframes/s'%(T,d/n*1e3,B*n*T/t/1e6))
    del x,y,eng
