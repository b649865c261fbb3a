"""BatchEngine — thin Python owner of one native ds_handle (B independent utterances on one GPU).

All signal processing happens in libdsenh.so's HIP kernels; this class only moves pointers."""
import ctypes
import threading
import weakref

import numpy as np

from . import _lib as L


class _PinnedPool:
    """Output arrays of the host-buffer calls out of page-locked blocks (ds_host_alloc = hipHostMalloc).  A fresh `np.empty` is fresh pages:
    the download has to fault each one in and pin it first (10 s per call at B = 1024 is 0.65 GB of output — 40 % of the call).  An array handed
    out here owns its block exclusively, like any new array; when the last reference to it (and to every view of it) is gone, a finalizer puts
    the block back, and the next call of that size gets pages that are resident and pinned already.  A caller who keeps every output just
    makes the pool allocate a new block per call.  Blocks of at least MIN_BYTES only (small outputs stay ordinary arrays); at most MAX_IDLE
    idle bytes are kept."""
    MIN_BYTES = 1 << 20
    MAX_IDLE = 4 << 30

    def __init__(self, lib):
        self._lib = lib
        self._free = {}                  # bytes -> [address, ...]
        self._idle = 0
        self._lock = threading.Lock()

    def empty(self, shape, dtype):
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dtype.itemsize
        if nbytes < self.MIN_BYTES:
            return np.empty(shape, dtype=dtype)
        with self._lock:
            lst = self._free.get(nbytes)
            addr = lst.pop() if lst else None
            if addr is not None:
                self._idle -= nbytes
        if addr is None:
            addr = self._lib.ds_host_alloc(nbytes)
            if not addr:
                return np.empty(shape, dtype=dtype)          # no page-locked memory left: an ordinary array
        block = (ctypes.c_char * nbytes).from_address(addr)
        weakref.finalize(block, self._give_back, addr, nbytes)     # the ctypes object is the base of the array and of all its views
        return np.frombuffer(block, dtype=dtype).reshape(shape)

    def _give_back(self, addr, nbytes):
        with self._lock:
            if self._idle + nbytes <= self.MAX_IDLE:
                self._free.setdefault(nbytes, []).append(addr)
                self._idle += nbytes
                return
        self._lib.ds_host_free(ctypes.c_void_p(addr))


_pool = None


def _pinned_pool():
    global _pool
    if _pool is None:
        _pool = _PinnedPool(L.load())
    return _pool


class BatchEngine:
    def __init__(self, algo, n_mics, nfft, hop=None, batch=1, track_ryy=False, mcra_L=15, device=-1,
                 alpha_y=0.0, alpha_v=0.0, diag=0.0, gate=0.0, mu=0.0, filter_len=0, no_norm=False, filt_mu=0.0,
                 filt_alpha=0.0, rls_lambda=0.0):
        self._lib = L.load()
        hop = nfft // 2 if hop is None else int(hop)
        cfg = L.ds_config(ctypes.sizeof(L.ds_config), int(algo), int(n_mics), int(nfft), hop, int(batch),
                          int(bool(track_ryy)), int(mcra_L), int(device), alpha_y, alpha_v, diag, gate, mu,
                          int(filter_len), int(bool(no_norm)), filt_mu, filt_alpha, rls_lambda)
        h = ctypes.c_void_p()
        L.check(self._lib.ds_create(ctypes.byref(cfg), ctypes.byref(h)))
        self._h = h
        self.algo, self.M, self.nfft, self.hop, self.batch = int(algo), int(n_mics), int(nfft), hop, int(batch)
        self.K = nfft // 2 + 1
        self.track_ryy = bool(track_ryy)

    # -- lifetime -------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.ds_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        L.check(self._lib.ds_reset(self._h), self._h)

    # -- set-up ---------------------------------------------------------------------------------
    def set_steering(self, a):
        """a: complex [K, M] (shared) or [B, K, M] (one look direction per utterance)."""
        a = np.ascontiguousarray(a, dtype=np.complex64)
        if a.shape[-2:] != (self.K, self.M) or a.ndim not in (2, 3) or (a.ndim == 3 and a.shape[0] != self.batch):
            raise ValueError("steering must be [K=%d, M=%d] or [B=%d, K, M], got %s" % (self.K, self.M, self.batch, a.shape))
        L.check(self._lib.ds_set_steering(self._h, a.ctypes.data_as(ctypes.c_void_p), int(a.ndim == 3)), self._h)

    def set_method(self, method):
        L.check(self._lib.ds_set_param_i(self._h, L.PARAM_METHOD, int(method)), self._h)

    def set_split(self, n):
        """fused frame kernels: n utterance groups as parallel hipGraph branches of process_device_seq(graph=1); DS_ALGO_WPE_MVDR chain:
        n utterance groups pipelined through the stages (default 4 from 256 utterances up, 1 = every stage over the whole batch)."""
        L.check(self._lib.ds_set_param_i(self._h, L.PARAM_SPLIT, int(n)), self._h)

    def set_wpe_delay(self, frames):
        """prediction delay (frames) of a DS_ALGO_WPE_MVDR chain handle; before the first call."""
        L.check(self._lib.ds_set_param_i(self._h, L.PARAM_WPE_DELAY, int(frames)), self._h)

    def set_window(self, window):
        """Transform handles: analysis / synthesis window [nfft] instead of the default sqrt-Hann (transform.py:415-419)."""
        w = np.ascontiguousarray(window, dtype=np.float32)
        L.check(self._lib.ds_set_window(self._h, self._p(w), int(w.size)), self._h)

    def set_mcspp_repeat(self, on):
        """McSpp handles: estimation(repeat=True), a second estimation_core after the noise update (mcspp.py:280-282)."""
        L.check(self._lib.ds_set_param_i(self._h, L.PARAM_MCSPP_REPEAT, int(bool(on))), self._h)

    def set_mcra_L(self, value):
        L.check(self._lib.ds_set_param_i(self._h, L.PARAM_MCRA_L, int(value)), self._h)

    def set_param_i(self, pid, value):
        L.check(self._lib.ds_set_param_i(self._h, int(pid), int(value)), self._h)

    def set_param_f(self, pid, value):
        L.check(self._lib.ds_set_param_f(self._h, int(pid), float(value)), self._h)

    # -- hot path -------------------------------------------------------------------------------
    def process(self, x, layout, out_dtype=np.float32):
        """Host arrays.  x: [B, L, M] (layout 0) or [B, M, L] (layout 1) -> y [B, L] float32 (or float64: widened on the device, ds_process_f64)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim != 3 or x.shape[0] != self.batch:
            raise ValueError("x must be [B=%d, ...] 3-D, got %s" % (self.batch, x.shape))
        n = x.shape[1] if layout == L.LAYOUT_SAMPLES_CHANNELS else x.shape[2]
        m = x.shape[2] if layout == L.LAYOUT_SAMPLES_CHANNELS else x.shape[1]
        if m != self.M:
            raise ValueError("expected %d channels, got %d" % (self.M, m))
        f64 = np.dtype(out_dtype) == np.float64
        y = _pinned_pool().empty((self.batch, n), np.float64 if f64 else np.float32)      # (a page-locked block from 1 MB up: see _PinnedPool)
        L.check((self._lib.ds_process_f64 if f64 else self._lib.ds_process)(self._h, x.ctypes.data_as(ctypes.c_void_p), int(layout), int(n),
                                                                           y.ctypes.data_as(ctypes.c_void_p)), self._h)
        return y

    def process_pcm16(self, pcm, first_channel=0):
        """Realtime wire format: pcm int16 [B, L, C_total] interleaved -> enhanced int16 [B, L] (conversion on the GPU)."""
        pcm = np.ascontiguousarray(pcm, dtype="<i2")
        if pcm.ndim != 3 or pcm.shape[0] != self.batch:
            raise ValueError("pcm must be [B=%d, samples, channels]" % self.batch)
        out = _pinned_pool().empty(pcm.shape[:2], "<i2")
        L.check(self._lib.ds_process_pcm16(self._h, pcm.ctypes.data_as(ctypes.c_void_p), int(pcm.shape[2]), int(first_channel),
                                           int(pcm.shape[1]), out.ctypes.data_as(ctypes.c_void_p)), self._h)
        return out

    def process_device(self, x_ptr, layout, x_batch_stride, n_samples, y_ptr, y_batch_stride, first=0, count=None,
                       stream=None, x_chan_stride=0):
        """Device pointers (ints), asynchronous on `stream` (int hipStream_t) or the handle's stream."""
        count = self.batch - first if count is None else count
        L.check(self._lib.ds_process_device(self._h, ctypes.c_void_p(x_ptr), int(layout), int(x_batch_stride),
                                            int(x_chan_stride), int(n_samples), ctypes.c_void_p(y_ptr),
                                            int(y_batch_stride), int(first),
                                            int(count), ctypes.c_void_p(stream) if stream else None), self._h)

    def process_device_seq(self, x_ptr, layout, x_batch_stride, x_chan_stride, x_call_stride, n_samples_per_call,
                           n_calls, y_ptr, y_batch_stride, y_call_stride, first=0, count=None, stream=None, graph=0):
        """n_calls successive process_device() calls enqueued by one native call (graph=1: hipGraph replay,
        graph=2: only build the graph)."""
        count = self.batch - first if count is None else count
        L.check(self._lib.ds_process_device_seq(
            self._h, ctypes.c_void_p(x_ptr), int(layout), int(x_batch_stride), int(x_chan_stride), int(x_call_stride),
            int(n_samples_per_call), int(n_calls), ctypes.c_void_p(y_ptr), int(y_batch_stride), int(y_call_stride),
            int(first), int(count), ctypes.c_void_p(stream) if stream else None, int(graph)), self._h)

    def synchronize(self):
        L.check(self._lib.ds_synchronize(self._h), self._h)

    def timing_begin(self):
        L.check(self._lib.ds_timing_begin(self._h), self._h)

    def timing_end(self):
        ms = ctypes.c_float(0)
        L.check(self._lib.ds_timing_end(self._h, ctypes.byref(ms)), self._h)
        return ms.value

    # -- frame-level objects (host arrays) -------------------------------------------------------
    @staticmethod
    def _p(a):
        return a.ctypes.data_as(ctypes.c_void_p)

    def stft(self, x, layout):
        """x [B, L, M] (layout 0) or [B, M, L] (layout 1) -> Y complex64 [B, T, K, M]."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        n = x.shape[1] if layout == L.LAYOUT_SAMPLES_CHANNELS else x.shape[2]
        Y = np.empty((self.batch, n // self.hop, self.K, self.M), dtype=np.complex64)
        L.check(self._lib.ds_stft(self._h, self._p(x), int(layout), int(n), self._p(Y), L.MEM_HOST), self._h)
        return Y

    def istft(self, Y):
        """Y complex [B, T, K, C] (C <= M) -> y float32 [B, T*hop, C]."""
        Y = np.ascontiguousarray(Y, dtype=np.complex64)
        B, T, K, C = Y.shape
        y = np.empty((B, T * self.hop, C), dtype=np.float32)
        L.check(self._lib.ds_istft(self._h, self._p(Y), int(T), int(C), self._p(y), L.MEM_HOST), self._h)
        return y

    def mcra_estimate(self, Y):
        """Y [B, T, K] float power or complex -> lambda_d [B, T, K]."""
        cplx = np.iscomplexobj(Y)
        Y = np.ascontiguousarray(Y, dtype=np.complex64 if cplx else np.float32)
        out = np.empty(Y.shape, dtype=np.float32)
        L.check(self._lib.ds_mcra_estimate(self._h, self._p(Y), int(cplx), int(Y.shape[1]), self._p(out), L.MEM_HOST), self._h)
        return out

    def mcra_estimate_p(self, Y):
        """Y [B, T, K] float power or complex -> (lambda_d, p) [B, T, K]: p = the speech presence probability after each frame."""
        cplx = np.iscomplexobj(Y)
        Y = np.ascontiguousarray(Y, dtype=np.complex64 if cplx else np.float32)
        lam = np.empty(Y.shape, dtype=np.float32)
        p = np.empty(Y.shape, dtype=np.float32)
        L.check(self._lib.ds_mcra_estimate_p(self._h, self._p(Y), int(cplx), int(Y.shape[1]), self._p(lam), self._p(p), L.MEM_HOST), self._h)
        return lam, p

    def mcmcra_estimate(self, y):
        """y complex [B, T, K, M] -> (p, G) [B, T, K]."""
        y = np.ascontiguousarray(y, dtype=np.complex64)
        p = np.empty(y.shape[:3], dtype=np.float32)
        G = np.empty(y.shape[:3], dtype=np.float32)
        L.check(self._lib.ds_mcmcra_estimate(self._h, self._p(y), int(y.shape[1]), self._p(p), self._p(G), L.MEM_HOST), self._h)
        return p, G

    def mcsppbase_estimate(self, y):
        """y complex [B, T, K, M] -> (p [B, T, K], w complex [B, T, K, M])."""
        y = np.ascontiguousarray(y, dtype=np.complex64)
        p = np.empty(y.shape[:3], dtype=np.float32)
        w = np.empty(y.shape, dtype=np.complex64)
        L.check(self._lib.ds_mcsppbase_estimate(self._h, self._p(y), int(y.shape[1]), self._p(p), self._p(w), L.MEM_HOST), self._h)
        return p, w

    def set_aux(self, table):
        t = np.ascontiguousarray(table, dtype=np.float32)
        L.check(self._lib.ds_set_aux(self._h, self._p(t), t.size), self._h)

    def mcspp_estimate(self, y, want_yout=True, want_matrices=False):
        """y complex [B, T, K, M] -> dict(p, w_pmwf[, yout][, phi_xx, phi_vv_inv])."""
        y = np.ascontiguousarray(y, dtype=np.complex64)
        B, T, K, M = y.shape
        out = {"p": np.empty((B, T, K), np.float32), "w_pmwf": np.empty((B, T, K, M), np.complex64)}
        if want_yout:
            out["yout"] = np.empty((B, T, K), np.complex64)
        if want_matrices:
            out["phi_xx"] = np.empty((B, T, K, M, M), np.complex64)
            out["phi_vv_inv"] = np.empty((B, T, K, M, M), np.complex64)
        g = lambda k: self._p(out[k]) if k in out else None
        L.check(self._lib.ds_mcspp_estimate(self._h, self._p(y), int(T), g("p"), g("w_pmwf"), g("yout"), g("phi_xx"),
                                            g("phi_vv_inv"), L.MEM_HOST), self._h)
        return out

    def steering(self, XX):
        """XX complex [B, K, M, M] -> principal eigenvectors [B, K, M] (phase-normalised by element 0)."""
        XX = np.ascontiguousarray(XX, dtype=np.complex64)
        v = np.empty(XX.shape[:3], dtype=np.complex64)
        L.check(self._lib.ds_steering(self._h, self._p(XX), self._p(v), L.MEM_HOST), self._h)
        return v

    def mvdr_weight(self, steer, Rinv):
        steer = np.ascontiguousarray(steer, dtype=np.complex64)
        Rinv = np.ascontiguousarray(Rinv, dtype=np.complex64)
        w = np.empty(steer.shape, dtype=np.complex64)
        L.check(self._lib.ds_mvdr_weight(self._h, self._p(steer), self._p(Rinv), self._p(w), L.MEM_HOST), self._h)
        return w

    def pmwf_weight(self, xi, Rxx, Rvv_inv, beta=1.0):
        xi = np.ascontiguousarray(xi, dtype=np.float32)
        Rxx = np.ascontiguousarray(Rxx, dtype=np.complex64)
        Rvv_inv = np.ascontiguousarray(Rvv_inv, dtype=np.complex64)
        w = np.empty(Rxx.shape[:3], dtype=np.complex64)
        L.check(self._lib.ds_pmwf_weight(self._h, self._p(xi), self._p(Rxx), self._p(Rvv_inv), float(beta), self._p(w), L.MEM_HOST), self._h)
        return w

    def gev_vector(self, target, noise):
        target = np.ascontiguousarray(target, dtype=np.complex64)
        noise = np.ascontiguousarray(noise, dtype=np.complex64)
        v = np.empty(target.shape[:3], dtype=np.complex64)
        L.check(self._lib.ds_gev_vector(self._h, self._p(target), self._p(noise), self._p(v), L.MEM_HOST), self._h)
        return v

    def blind_analytic_normalization(self, vector, noise, eps=0.0):
        vector = np.ascontiguousarray(vector, dtype=np.complex64)
        noise = np.ascontiguousarray(noise, dtype=np.complex64)
        out = np.empty(vector.shape, dtype=np.complex64)
        L.check(self._lib.ds_blind_analytic_normalization(self._h, self._p(vector), self._p(noise), float(eps), self._p(out), L.MEM_HOST), self._h)
        return out

    def phase_correction(self, vector):
        vector = np.ascontiguousarray(vector, dtype=np.complex64)
        out = np.empty(vector.shape, dtype=np.complex64)
        L.check(self._lib.ds_phase_correction(self._h, self._p(vector), self._p(out), L.MEM_HOST), self._h)
        return out

    def tdfilter_update(self, x, d, p=1.0):
        """x, d [B, n] samples -> err [B, n]  (n successive sample-wise NLMS / RLS updates)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        d = np.ascontiguousarray(d, dtype=np.float32)
        err = np.empty_like(x)
        L.check(self._lib.ds_tdfilter_update(self._h, self._p(x), self._p(d), int(x.shape[1]), float(p), self._p(err),
                                             L.MEM_HOST), self._h)
        return err

    def chain_set_aux(self, which, table):
        """constant table of a chain handle (L.CHAIN_AUX_FIR: TimeAlignment coefficients [L, M]; L.CHAIN_AUX_COHERENCE: Fn [K])."""
        t = np.ascontiguousarray(table, dtype=np.float32)
        L.check(self._lib.ds_chain_set_aux(self._h, int(which), self._p(t), t.size), self._h)

    def mcspp_mvdr_process(self, x, layout, want_p=True):
        """DS_ALGO_MCSPP_MVDR: x [B, n, M] (layout 0) or [B, M, n] (layout 1) -> (y [B, n], p [B, T, K] or None)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim != 3:
            raise ValueError("x must be [B=%d, n, M=%d] or [B, M, n]" % (self.batch, self.M))
        n = x.shape[1] if layout == L.LAYOUT_SAMPLES_CHANNELS else x.shape[2]
        m = x.shape[2] if layout == L.LAYOUT_SAMPLES_CHANNELS else x.shape[1]
        if x.shape[0] != self.batch or m != self.M:
            raise ValueError("x must be [B=%d, n, M=%d] or [B, M, n]" % (self.batch, self.M))
        if n % self.hop != 0:
            raise ValueError("n_samples (%d) must be a multiple of hop (%d)" % (n, self.hop))
        y = np.empty((self.batch, n), dtype=np.float32)
        p = np.empty((self.batch, n // self.hop, self.K), dtype=np.float32) if want_p else None
        L.check(self._lib.ds_mcspp_mvdr_process(self._h, self._p(x), int(layout), int(n), self._p(y), self._p(p) if want_p else None, L.MEM_HOST), self._h)
        return y, p

    def subband_gsc_process(self, x, extras=True):
        """DS_ALGO_SUBBAND_GSC: x [B, M, n] -> (y [B, n], fix_output [B, n], bm_output [B, M, n], p [B, T, K], aligned [B, M, n])
        (the four extras are None with extras=False)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim != 3 or x.shape[0] != self.batch or x.shape[1] != self.M:
            raise ValueError("x must be [B=%d, M=%d, n]" % (self.batch, self.M))
        n = x.shape[2]
        y = np.empty((self.batch, n), dtype=np.float32)
        fix = bm = p = al = None
        if extras:
            fix = np.empty((self.batch, n), dtype=np.float32)
            bm = np.empty((self.batch, self.M, n), dtype=np.float32)
            p = np.empty((self.batch, n // self.hop, self.K), dtype=np.float32)
            al = np.empty((self.batch, self.M, n), dtype=np.float32)
        q = lambda a: self._p(a) if a is not None else None
        L.check(self._lib.ds_subband_gsc_process(self._h, self._p(x), int(n), self._p(y), q(fix), q(bm), q(p), q(al), L.MEM_HOST), self._h)
        return y, fix, bm, p, al

    def tdgsc_process(self, x, postfilter=False):
        """DS_ALGO_TDGSC: x [B, M, n] -> (out [B, n], p [B, T, K], bm [B, n, M-1], w [B, frameLen, M-1])."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim != 3 or x.shape[0] != self.batch or x.shape[1] != self.M:
            raise ValueError("x must be [B=%d, M=%d, n]" % (self.batch, self.M))
        n = x.shape[2]
        out = np.empty((self.batch, n), dtype=np.float32)
        p = np.empty((self.batch, n // self.hop, self.K), dtype=np.float32)
        bm = np.empty((self.batch, n, self.M - 1), dtype=np.float32)
        w = np.empty((self.batch, self.hop, self.M - 1), dtype=np.float32)
        L.check(self._lib.ds_tdgsc_process(self._h, self._p(x), int(n), int(bool(postfilter)), self._p(out), self._p(p), self._p(bm), self._p(w),
                                           L.MEM_HOST), self._h)
        return out, p, bm, w

    def fdgsc_process(self, x, postfilter=False, dc_notch=True):
        """DS_ALGO_FDGSC: x [B, M, n] -> dict(out [B, n], p [B, T, K], fix, fix_d [B, n], bm, al, al_d [B, M, n], w_aic [B, frameLen, M],
        w_bm [B * M, frameLen])."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim != 3 or x.shape[0] != self.batch or x.shape[1] != self.M:
            raise ValueError("x must be [B=%d, M=%d, n]" % (self.batch, self.M))
        B, M, n = x.shape
        f32 = np.float32
        r = dict(out=np.empty((B, n), f32), p=np.empty((B, n // self.hop, self.K), f32), fix=np.empty((B, n), f32), fix_d=np.empty((B, n), f32),
                 bm=np.empty((B, M, n), f32), al=np.empty((B, M, n), f32), al_d=np.empty((B, M, n), f32),
                 w_aic=np.empty((B, self.hop, M), f32), w_bm=np.empty((B * M, self.hop), f32))
        L.check(self._lib.ds_fdgsc_process(self._h, self._p(x), int(n), int(bool(postfilter)), int(bool(dc_notch)), *[self._p(r[k]) for k in
                                           ("out", "p", "fix", "fix_d", "bm", "al", "al_d", "w_aic", "w_bm")], L.MEM_HOST), self._h)
        return r

    def adaptive_frames(self, Z, gain=None):
        """Z complex [B, T, K, M] STFT frames, gain [B, T, K] or None -> Y complex [B, T, K]: the adaptivebeamfomer frame loop
        (MCRA-gated Rvv, src/DS/MVDR weights) as a frame-level operator, times the optional post-filter gain."""
        Z = np.ascontiguousarray(Z, dtype=np.complex64)
        g = None if gain is None else np.ascontiguousarray(gain, dtype=np.float32)
        Y = np.empty(Z.shape[:3], dtype=np.complex64)
        L.check(self._lib.ds_adaptive_frames(self._h, self._p(Z), self._p(g) if g is not None else None, int(Z.shape[1]),
                                             self._p(Y), L.MEM_HOST), self._h)
        return Y

    def set_fdaf(self, kind=0, constrain=True, non_causal=False, weight_norm=False, two_path=False):
        """select the overlap-save FDAF variant of a DS_ALGO_FDAF handle (plain / clamped blocking filter / norm-limited canceller)."""
        for pid, v in ((L.PARAM_FDAF_KIND, kind), (L.PARAM_FDAF_CONSTRAIN, constrain), (L.PARAM_FDAF_NON_CAUSAL, non_causal),
                       (L.PARAM_FDAF_WEIGHT_NORM, weight_norm), (L.PARAM_FDAF_TWO_PATH, two_path)):
            L.check(self._lib.ds_set_param_i(self._h, pid, int(v)), self._h)

    def fdaf_update(self, x, d, p=None, fir_truncate=None, want_w=True, p_complement=False):
        """x [B, T*L, C], d [B, T*L] samples (L = nfft/2), p None | [B, T] | [B, T, K] -> (err [B, T*L], w [B, L, C] | None):
        T successive block updates in one launch."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        d = np.ascontiguousarray(d, dtype=np.float32)
        Lb = self.nfft // 2
        if x.ndim != 3 or x.shape[0] != self.batch or x.shape[2] != self.M or x.shape[1] % Lb or d.shape != x.shape[:2]:
            raise ValueError("fdaf_update: x must be [B=%d, T*%d, C=%d] and d [B, T*%d]" % (self.batch, Lb, self.M, Lb))
        T = x.shape[1] // Lb
        if p is None:
            pm, pp = L.FDAF_P_NONE, None
        else:
            pp = np.ascontiguousarray(p, dtype=np.float32)
            if pp.shape == (self.batch, T):
                pm = L.FDAF_P_BLOCK
            elif pp.shape == (self.batch, T, self.K):
                pm = L.FDAF_P_BIN
            else:
                raise ValueError("fdaf_update: p must be [B, T] or [B, T, K=%d], got %s" % (self.K, pp.shape))
        if p_complement:
            pm |= L.FDAF_P_COMPLEMENT                    # the kernel forms 1 - p
        err = np.empty_like(d)
        w = np.empty((self.batch, Lb, self.M), dtype=np.float32) if want_w else None
        L.check(self._lib.ds_fdaf_update(self._h, self._p(x), self._p(d), self._p(pp) if pp is not None else None, pm, T,
                                         -1 if fir_truncate is None else int(fir_truncate), self._p(err),
                                         self._p(w) if want_w else None, L.MEM_HOST), self._h)
        return err, w

    def fdaf_state(self):
        """(W complex [B, C, K], P [B, K]) of a DS_ALGO_FDAF handle."""
        nbytes = self._lib.ds_field_bytes(self._h, L.FIELD_OP_STATE)
        out = np.empty(nbytes // 4, dtype=np.float32)
        L.check(self._lib.ds_get_state(self._h, L.FIELD_OP_STATE, out.ctypes.data_as(ctypes.c_void_p), nbytes), self._h)
        st = out.reshape(self.batch, -1)
        C, K = self.M, self.K
        W = st[:, : 2 * C * K].copy().view(np.complex64).reshape(self.batch, C, K)
        return W, st[:, 2 * C * K: 2 * C * K + K].copy()

    def fdaf_foreground(self):
        """foreground filter [B, C, K] complex of a two-path DS_ALGO_FDAF handle (FastFreqLms.foreground)."""
        nbytes = self._lib.ds_field_bytes(self._h, L.FIELD_OP_STATE)
        out = np.empty(nbytes // 4, dtype=np.float32)
        L.check(self._lib.ds_get_state(self._h, L.FIELD_OP_STATE, out.ctypes.data_as(ctypes.c_void_p), nbytes), self._h)
        st = out.reshape(self.batch, -1)
        C, K, Lb = self.M, self.K, self.nfft // 2
        o = 2 * C * K + K + C * Lb + Lb // 2
        return st[:, o: o + 2 * C * K].copy().view(np.complex64).reshape(self.batch, C, K)

    def tdfilter_weights(self):
        nbytes = self._lib.ds_field_bytes(self._h, L.FIELD_OP_STATE)
        out = np.empty(nbytes // 4, dtype=np.float32)
        L.check(self._lib.ds_get_state(self._h, L.FIELD_OP_STATE, out.ctypes.data_as(ctypes.c_void_p), nbytes), self._h)
        return out.reshape(self.batch, -1)

    def dcnotch(self, x):
        """x [B, M, n] float32 -> y [B, M, n]  (FilterDcNotch16 per channel, state carried)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.empty_like(x)
        L.check(self._lib.ds_dcnotch(self._h, self._p(x), int(x.shape[2]), self._p(y), L.MEM_HOST), self._h)
        return y

    def firbank(self, x, want_mean=True, want_bm=False):
        """x [B, n, M] -> (y [B, n, M], channel mean [B, n])  (TimeAlignment FIR bank, history carried);
        want_bm adds the adjacent-pair differences y[m] - y[m+1] as a third result [B, n, M-1] (TDGSC.py:69-87)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.empty_like(x)
        mean = np.empty(x.shape[:2], dtype=np.float32) if want_mean else None
        if not want_bm:
            L.check(self._lib.ds_firbank(self._h, self._p(x), int(x.shape[1]), self._p(y), self._p(mean) if want_mean else None,
                                         L.MEM_HOST), self._h)
            return y, mean
        bm = np.empty(x.shape[:2] + (x.shape[2] - 1,), dtype=np.float32)
        L.check(self._lib.ds_firbank_bm(self._h, self._p(x), int(x.shape[1]), self._p(y), self._p(mean) if want_mean else None,
                                        self._p(bm), L.MEM_HOST), self._h)
        return y, mean, bm

    def omlsa_estimate(self, y, u):
        """y [B, T, K], u [B, T, K, M-1] powers -> (lambda_d, G, p) [B, T, K]."""
        y = np.ascontiguousarray(y, dtype=np.float32)
        u = np.ascontiguousarray(u, dtype=np.float32)
        lam, G, p = (np.empty(y.shape, dtype=np.float32) for _ in range(3))
        L.check(self._lib.ds_omlsa_estimate(self._h, self._p(y), self._p(u), int(y.shape[1]), self._p(lam), self._p(G),
                                            self._p(p), L.MEM_HOST), self._h)
        return lam, G, p

    def omlsa_postfilter(self, Y, U):
        """Y complex [B, T, K] beam spectrum, U complex [B, T, K, M-1] reference spectra -> (G [B, T, K], Y * sqrt(G) complex [B, T, K])."""
        Y = np.ascontiguousarray(Y, dtype=np.complex64)
        U = np.ascontiguousarray(U, dtype=np.complex64)
        G = np.empty(Y.shape, dtype=np.float32)
        Yout = np.empty(Y.shape, dtype=np.complex64)
        L.check(self._lib.ds_omlsa_postfilter(self._h, self._p(Y), self._p(U), int(Y.shape[1]), self._p(G), self._p(Yout), L.MEM_HOST), self._h)
        return G, Yout

    def sublms_update(self, x, d, p=None):
        """x complex [B, T, K, C], d complex [B, T, K], p [B, T, K] or None -> err complex [B, T, K]."""
        x = np.ascontiguousarray(x, dtype=np.complex64)
        d = np.ascontiguousarray(d, dtype=np.complex64)
        pp = None if p is None else np.ascontiguousarray(p, dtype=np.float32)
        err = np.empty(d.shape, dtype=np.complex64)
        L.check(self._lib.ds_sublms_update(self._h, self._p(x), self._p(d), self._p(pp) if pp is not None else None,
                                           int(d.shape[1]), self._p(err), L.MEM_HOST), self._h)
        return err

    def subrls_update(self, x, d):
        x = np.ascontiguousarray(x, dtype=np.complex64)
        d = np.ascontiguousarray(d, dtype=np.complex64)
        err = np.empty(d.shape, dtype=np.complex64)
        L.check(self._lib.ds_subrls_update(self._h, self._p(x), self._p(d), int(d.shape[1]), self._p(err), L.MEM_HOST), self._h)
        return err

    def wpe_update(self, xd, d):
        """xd, d complex [B, T, K, C] -> err complex [B, T, K, C]."""
        xd = np.ascontiguousarray(xd, dtype=np.complex64)
        d = np.ascontiguousarray(d, dtype=np.complex64)
        err = np.empty(d.shape, dtype=np.complex64)
        L.check(self._lib.ds_wpe_update(self._h, self._p(xd), self._p(d), int(d.shape[1]), self._p(err), L.MEM_HOST), self._h)
        return err

    def op_state_raw(self):
        """the operator state exactly as it sits in HBM (flat float32 per utterance)."""
        nbytes = self._lib.ds_field_bytes(self._h, L.FIELD_OP_STATE)
        out = np.empty(nbytes // 4, dtype=np.float32)
        L.check(self._lib.ds_get_state(self._h, L.FIELD_OP_STATE, out.ctypes.data_as(ctypes.c_void_p), nbytes), self._h)
        return out.reshape(self.batch, -1)

    def op_state(self):
        """raw operator state [B, NF, K] (rows documented in distantspeech_amd/ops.py)."""
        nbytes = self._lib.ds_field_bytes(self._h, L.FIELD_OP_STATE)
        out = np.empty(nbytes // 4, dtype=np.float32)
        L.check(self._lib.ds_get_state(self._h, L.FIELD_OP_STATE, out.ctypes.data_as(ctypes.c_void_p), nbytes), self._h)
        KP = L.plane_len(self.K)
        # in HBM the operator kernels keep float f of bin k at [f // 4][k][f % 4] (float4 planes: 16-byte accesses per lane)
        return out.reshape(self.batch, -1, KP, 4).transpose(0, 1, 3, 2).reshape(self.batch, -1, KP)[:, :, : self.K]

    # -- state ----------------------------------------------------------------------------------
    def get_field(self, field):
        nbytes = self._lib.ds_field_bytes(self._h, int(field))
        if nbytes == 0:
            raise AttributeError("state field %d is not available for this configuration" % field)
        B, K, M = self.batch, self.K, self.M
        if field == L.FIELD_COUNTERS:
            out = np.empty((B, 4), dtype=np.int32)
        else:
            out = np.empty(nbytes // 4, dtype=np.float32)
        L.check(self._lib.ds_get_state(self._h, int(field), out.ctypes.data_as(ctypes.c_void_p), nbytes), self._h)
        if field in (L.FIELD_RVV, L.FIELD_RYY):
            return out.view(np.complex64).reshape(B, K, M, M)
        if field in (L.FIELD_PHI_YY, L.FIELD_PHI_VV):
            return out.reshape(B, K, M, M)
        if field == L.FIELD_G_AIC:
            return out.view(np.complex64).reshape(B, K, M - 1)
        if field == L.FIELD_H:
            return out.view(np.complex64).reshape(B, K, M)
        if field == L.FIELD_REF_POWERS:
            return out.reshape(B, -1, K, M)                 # [B, T of the last call, K, M]
        if field == L.FIELD_STFT_TAIL:
            return out.reshape(B, M, self.nfft - self.hop)
        if field == L.FIELD_OLA_TAIL:
            return out.reshape(B, M, self.nfft - self.hop) if self.algo == L.ALGO_TRANSFORM else out.reshape(B, self.hop)
        if field == L.FIELD_COUNTERS:
            return out
        if field == L.FIELD_NOTCH_MEM:
            return out.reshape(B, M, 2)
        return out.reshape(B, K)

    def state_bytes(self):
        """bytes of the carried state as the library packs it (without the checkpoint's framing): what one call reads and writes back"""
        return int(self._lib.ds_state_payload_bytes(self._h))

    def chain_stages(self):
        """[(stage index, DS_ALGO_*, channels, batch, carried-state bytes)] of a chain handle (ds_chain_stage_info); [] for a plain handle"""
        out = []
        a, m, b, n = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_size_t()
        for i in range(10):
            if self._lib.ds_chain_stage_info(self._h, i, ctypes.byref(a), ctypes.byref(m), ctypes.byref(b), ctypes.byref(n)) == 0:
                out.append((i, a.value, m.value, b.value, int(n.value)))
        return out

    def stage_state_raw(self, i, field=None):
        """state field of stage i of a chain handle exactly as it sits in HBM (flat float32 per utterance): ds_chain_stage_state"""
        field = L.FIELD_OP_STATE if field is None else int(field)
        nbytes = self._lib.ds_chain_stage_field_bytes(self._h, int(i), field)
        if nbytes == 0:
            raise AttributeError("stage %d has no state field %d" % (i, field))
        out = np.empty(nbytes // 4, dtype=np.float32)
        L.check(self._lib.ds_chain_stage_state(self._h, int(i), field, out.ctypes.data_as(ctypes.c_void_p), nbytes), self._h)
        return out.reshape(self.batch, -1)

    def stage_state_f64(self, i, field):
        """a float64 state field of stage i of a chain handle (DS_FIELD_WPE_STATE64), flat per utterance"""
        nbytes = self._lib.ds_chain_stage_field_bytes(self._h, int(i), int(field))
        if nbytes == 0:
            raise AttributeError("stage %d has no state field %d" % (i, field))
        out = np.empty(nbytes // 8, dtype=np.float64)
        L.check(self._lib.ds_chain_stage_state(self._h, int(i), int(field), out.ctypes.data_as(ctypes.c_void_p), nbytes), self._h)
        return out.reshape(self.batch, -1)

    def export_state(self):
        n = self._lib.ds_state_bytes(self._h)
        buf = np.empty(n, dtype=np.uint8)
        L.check(self._lib.ds_export_state(self._h, buf.ctypes.data_as(ctypes.c_void_p), n), self._h)
        return buf

    def import_state(self, buf):
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        L.check(self._lib.ds_import_state(self._h, buf.ctypes.data_as(ctypes.c_void_p), buf.size), self._h)
