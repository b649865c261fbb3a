import sys, time, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/scratch')
from wpe_sample import stream_spectra, rms, C, NT
D, Xd = stream_spectra(11)                    # [T, 17, C] complex64
T, K, _ = D.shape; CN = C*NT
lam = 0.998
def run(mode):
    P = np.tile(np.eye(CN, dtype=complex)*1e-3, (K,1,1)); W = np.zeros((K,C,CN),complex); buf=np.zeros((K,C,NT),complex); var=np.zeros((K,1))
    out=np.zeros((T,K,C),complex)
    for t in range(T):
        buf[:,:,1:] = buf[:,:,:-1].copy(); buf[:,:,0]=Xd[t]
        X = buf.reshape(K,-1)
        d = D[t].astype(complex)
        err = d - np.einsum('kmi,ki->km', W.conj(), X)
        var = 0.98*var + 0.02*(np.abs(np.einsum('ij,ij->i', d.conj(), d))/C)[:,None]
        g = np.einsum('kij,kj->ki', P, X)
        den = lam*var + np.sum(X.conj()*g,axis=-1,keepdims=True).real
        kn = g/den
        P = (P - g[:,:,None]*np.conj(g[:,None,:])/den[:,:,None])/lam      # Hermitian-preserving form (the kernels')
        if mode != 'f64':
            Pr = P.astype(np.complex64).astype(complex)
            if mode == 'diag64':
                idx=np.arange(CN); Pr[:,idx,idx] = P[:,idx,idx].real
            if mode == 'diag_ff':      # diagonal as float-float (hi + lo fp32 pair)
                idx=np.arange(CN); dd = P[:,idx,idx].real; hi=dd.astype(np.float32).astype(float); lo=(dd-hi).astype(np.float32).astype(float); Pr[:,idx,idx]=hi+lo
            P = Pr
        for ch in range(C): W[:,ch,:] += err[:,ch:ch+1].conj()*kn
        out[t]=err
    return out
t0=time.time(); ref=run('f64'); print("f64 %.0f s"%(time.time()-t0), flush=True)
for mode in ('f32','diag64','diag_ff'):
    o=run(mode)
    rel=[rms(o[a:a+250]-ref[a:a+250])/rms(ref[a:a+250]) for a in range(0,T,250)]
    print(mode, "worst %.2e last %.2e" % (max(rel), rel[-1]), flush=True)
