"""ctypes wrapper of the serial CPU execution of the kernel block program (TEST INFRASTRUCTURE).

Builds tests/emul/libds_emul.so from tests/emul/ds_emul.cpp + distantspeech_amd/csrc/ds_core.hpp with
g++ on first use.  Used only by tests/test_kernel_emul.py to check the kernel's arithmetic against
the oracle in the CPU-only container; never imported by the product package."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SO = os.path.join(HERE, "libds_emul.so")
SRCS = [os.path.join(HERE, "ds_emul.cpp"), os.path.join(ROOT, "distantspeech_amd", "csrc", "ds_core.hpp"),
        os.path.join(ROOT, "distantspeech_amd", "csrc", "ds_ops.hpp"), os.path.join(ROOT, "distantspeech_amd", "csrc", "ds_tables.hpp"),
        os.path.join(ROOT, "distantspeech_amd", "csrc", "ds_tdfilter.hpp"),
        os.path.join(ROOT, "distantspeech_amd", "csrc", "ds_fdaf.hpp"), os.path.join(ROOT, "distantspeech_amd", "csrc", "ds_wpe.hpp"), os.path.join(ROOT, "distantspeech_amd", "csrc", "ds_wpe2.hpp"), os.path.join(ROOT, "distantspeech_amd", "csrc", "ds_wpe_wide.hpp"),
        os.path.join(ROOT, "distantspeech_amd", "csrc", "ds_linalg64.hpp"), os.path.join(ROOT, "distantspeech_amd", "csrc", "ds_quad.hpp"),
        os.path.join(ROOT, "distantspeech_amd", "csrc", "ds_pipe.hpp")]


def build(force=False):
    if not force and os.path.exists(SO) and all(os.path.getmtime(SO) >= os.path.getmtime(s) for s in SRCS):
        return SO
    # -fno-strict-aliasing: the block programs fill LDS structures sixteen bytes at a time (vec4 stores over cf / float arrays) and read
    # them back by element in a LATER phase; on the device a workgroup barrier (with its fence) lies between the two, here the phases
    # are inlined loops and g++ -O2 has been seen moving the element loads above the vec4 stores (StftEngine<512, 5, false, 4>: all-zero output)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-mfma", "-fno-strict-aliasing", SRCS[0], "-o", SO])
    return SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.emul_layout.restype = ctypes.c_int
        _lib.emul_run.restype = ctypes.c_int
    return _lib


def set_pipe(on):
    """512-point frame programs as the hop-pipelined engine (ds_pipe.hpp) instead of the phase-by-phase one (ds_core.hpp)."""
    lib().emul_set_pipe(int(bool(on)))


class EmulEngine:
    """Holds the per-utterance state arrays the GPU keeps in HBM and runs the block program on them."""

    def __init__(self, algo, nfft, M, batch=1, ryy=False, mcra_L=15):
        self.algo, self.nfft, self.M, self.batch, self.ryy = algo, nfft, M, batch, int(ryy)
        self.hop, self.K = nfft // 2, nfft // 2 + 1
        kp = ctypes.c_int(0)
        self.NF = lib().emul_layout(algo, nfft, M, self.ryy, ctypes.byref(kp))
        self.KP = kp.value
        self.ust = (max(self.NF, 1) * self.KP + 31) & ~31                          # StateLayout::ust(): floats between utterances
        self.bins = np.zeros((batch, self.ust), dtype=np.float32)                    # per utterance: NF // 4 float4 planes [KP], then [KP][NF % 4]
        self.tail_in = np.zeros((batch, M, self.hop), dtype=np.float32)
        self.tail_out = np.zeros((batch, self.hop), dtype=np.float32)
        self.counters = np.zeros((batch, 4), dtype=np.int32)
        self.counters[:, 1] = 1
        self.steer = None
        self.method, self.mcra_L = 2, mcra_L
        self.alpha_y, self.alpha_v, self.diag, self.gate, self.mu = 0.8, 0.9998, 1e-6, 0.4, 0.01

    def set_steering(self, a):
        a = np.ascontiguousarray(a, dtype=np.complex64)
        self.steer_per_utt = int(a.ndim == 3)
        self.steer = a

    def process(self, x, layout, ref_pow=False):
        """x [B, L, M] (layout 0) or [B, M, L] (layout 1) float32 -> y [B, L]; ref_pow (GSC): also self.ref_pow [B, T, K, M], the powers of
        the canceller output and of the blocking-matrix outputs per frame and bin (Params::ref_pow)"""
        x = np.ascontiguousarray(x, dtype=np.float32)
        L = x.shape[1] if layout == 0 else x.shape[2]
        y = np.zeros((self.batch, L), dtype=np.float32)
        f = ctypes.c_float
        self.ref_pow = np.zeros((self.batch, L // self.hop, self.K, self.M), dtype=np.float32) if ref_pow else None
        lib().emul_set_ref_pow(_vp(self.ref_pow))
        rc = lib().emul_run(self.algo, self.nfft, self.M, self.ryy, self.batch, x.ctypes.data_as(ctypes.c_void_p), layout, L,
                            y.ctypes.data_as(ctypes.c_void_p), self.bins.ctypes.data_as(ctypes.c_void_p),
                            self.tail_in.ctypes.data_as(ctypes.c_void_p), self.tail_out.ctypes.data_as(ctypes.c_void_p),
                            self.counters.ctypes.data_as(ctypes.c_void_p), self.steer.ctypes.data_as(ctypes.c_void_p),
                            self.steer_per_utt, self.method, self.mcra_L, f(self.alpha_y), f(self.alpha_v), f(self.diag),
                            f(self.gate), f(self.mu))
        lib().emul_set_ref_pow(None)
        assert rc == 0, rc
        return y

    def field(self, f):
        """float index f of every bin -> [B, K]"""
        npf, rt = self.NF // 4, self.NF % 4
        if f < 4 * npf:
            return self.bins[:, : npf * self.KP * 4].reshape(self.batch, npf, self.KP, 4)[:, f // 4, : self.K, f % 4]
        return self.bins[:, npf * self.KP * 4: self.NF * self.KP].reshape(self.batch, self.KP, rt)[:, : self.K, f - 4 * npf]


def _vp(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


class EmulTransform:
    """Transform.stft / istft through the kernel block programs (StftEngine / IstftEngine)."""

    def __init__(self, nfft, M, batch=1, hop=None):
        self.nfft, self.M, self.batch, self.hop, self.K = nfft, M, batch, hop or nfft // 2, nfft // 2 + 1
        self.ov = nfft // self.hop
        self.tail_in = np.zeros((batch, M, nfft - self.hop), dtype=np.float32)
        self.tail_out = np.zeros((batch, M, nfft - self.hop), dtype=np.float32)

    def stft(self, x, layout=0):
        x = np.ascontiguousarray(x, dtype=np.float32)
        n = x.shape[1] if layout == 0 else x.shape[2]
        Y = np.zeros((self.batch, n // self.hop, self.K, self.M), dtype=np.complex64)
        assert lib().emul_stft_ov(self.nfft, self.ov, self.M, self.batch, _vp(x), layout, n, _vp(Y), _vp(self.tail_in)) == 0
        return Y

    def istft(self, Y):
        Y = np.ascontiguousarray(Y, dtype=np.complex64)
        B, T, K, C = Y.shape
        y = np.zeros((B, T * self.hop, C), dtype=np.float32)
        assert lib().emul_istft_ov(self.nfft, self.ov, self.M, self.batch, _vp(Y), T, C, _vp(y), _vp(self.tail_out)) == 0
        return y


class EmulOp:
    """One frame-level operator handle (state + uniform counters), mirrors run_binop() in ds_api.hip."""
    OPS = {"mcra": 0, "mcmcra": 1, "omlsa": 2, "sublms": 3, "subrls": 4, "mcsppbase": 5, "wpe": 6, "mccdr": 7, "mcspp": 8, "steering": 9, "mvdrw": 10,
           "pmwfw": 14, "gev": 15, "ban": 16, "phasecorr": 17}

    def __init__(self, op, nfft, M=1, N=2, batch=1, mu=None, alpha=0.9, lam=0.998, norm=1, L=15):
        self.op, self.B, self.K, self.M, self.N = self.OPS[op], batch, nfft // 2 + 1, M, N
        self.KP = (self.K + 7) & ~7                       # ds_core.hpp plane_len()
        self.NF = {0: 5, 1: M * (M + 1) + 4, 2: 5 * M + (M - 1) + 8, 3: 4 * N * M + 1, 4: 4 * N + 2 * N * N,
                   5: 2 * M * M + 8 + 2 * M,
                   6: 2 * M * M * N + 2 * M * N + 2 * (M * N) ** 2 + 1, 7: 9, 8: 12 + 2 * M * M + 3, 9: 1, 10: 1, 14: 1, 15: 1, 16: 1, 17: 1}[self.op]
        self.st = np.zeros((batch, self.NF, self.KP), dtype=np.float32)
        if self.op == 2:
            o = 5 * M + 1 + (M - 1)
            for f in (o + 1, o + 2, o + 3, o + 5, o + 6, 5 * M):
                self.st[:, f, :] = 1.0
        if self.op == 6:
            CN = M * N
            for i in range(CN):
                self.st[:, 2 * M * CN + 2 * CN + 2 * (i * CN + i), :] = 1e-3
        if self.op == 4:
            for i in range(N):
                self.st[:, 4 * N + 2 * (i * N + i), :] = 1000.0
        self.frm, self.ell, self.first, self.L = 0, 1, 1, L
        self.mu = mu if mu is not None else (0.5 if self.op == 4 else 0.1)
        self.alpha, self.lam, self.norm = alpha, lam, norm
        self.reg = 1e-4

    def run(self, in0, in1=None, in2=None, n_out=1, out_complex=False, in_complex=0, out_shapes=None):
        in0 = np.ascontiguousarray(in0)
        T = in0.shape[1]
        if out_shapes is not None:
            outs = [np.zeros((self.B, T, self.K) + tuple(sh), dtype=dt) for sh, dt in out_shapes]
            n_out = len(outs)
        else:
            outs = [np.zeros((self.B, T, self.K), dtype=np.complex64 if out_complex else np.float32) for _ in range(n_out)]
        o = outs + [None] * (5 - n_out)
        f = ctypes.c_float
        L = getattr(self, "L_override", self.L)
        rc = lib().emul_op(getattr(self, "op_override", self.op), self.B, self.K, T, _vp(self.st), self.NF, _vp(in0), _vp(in1),
                           _vp(in2), _vp(o[0]), _vp(o[1]), _vp(o[2]), _vp(o[3]), _vp(o[4]), self.M, getattr(self, "N_override", self.N),
                           self.frm, self.ell, L, self.first, int(in_complex), int(in2 is not None), self.norm, f(self.mu),
                           f(self.alpha), f(self.reg), f(self.lam))
        assert rc == 0
        if not getattr(self, "hold_counters", False):
            for _ in range(T):
                if self.frm != 0 and self.ell % L == 0:
                    self.ell = 0
                self.frm += 1
                self.ell += 1
            self.first = 0
        return outs

    def run_mcspp(self, y, Fn, want_yout=True, want_matrices=False, variant=8, repeat=False):
        """McSpp handle: McCDR pass (L = 65) then McSpp pass, like ds_mcspp_estimate().  variant: 8 = OP_MCSPP (all outputs),
        12 = OP_MCSPP_LEAN, 13 = OP_MCSPP_STEADY (p only)."""
        y = np.ascontiguousarray(y, dtype=np.complex64)
        B, T, K, M = y.shape
        self.op_override, self.L_override, self.hold_counters = 7, 65, True
        gamma = self.run(y, np.ascontiguousarray(Fn, dtype=np.float32))[0]
        self.op_override, self.N_override, self.hold_counters = variant, 9, False
        if variant != 8:
            return self.run(y, gamma, out_shapes=[((), np.float32)])
        shapes = [((), np.float32), ((M,), np.complex64)]
        shapes.append(((), np.complex64))
        if want_matrices:
            shapes += [((M, M), np.complex64), ((M, M), np.complex64)]
        lib().emul_set_repeat(int(repeat))
        try:
            outs = self.run(y, gamma, out_shapes=shapes)
        finally:
            lib().emul_set_repeat(0)
        return outs


class EmulFrontend:
    """FilterDcNotch16 + TimeAlignment FIR bank through the kernel per-thread programs (td_dcnotch / td_fir)."""

    def __init__(self, M, coef=None, radius=0.9, batch=1):
        self.M, self.B, self.radius = M, batch, radius
        self.mem = np.zeros((batch, M, 2), dtype=np.float64)             # the recursion and its memory run in double (ds_ops.hpp td_dcnotch)
        self.coef = None if coef is None else np.ascontiguousarray(coef, dtype=np.float32)
        if coef is not None:
            self.L = self.coef.shape[0]
            self.cache = [np.zeros((batch, M, self.L - 1), dtype=np.float32) for _ in range(2)]
            self.cur = 0

    def dcnotch(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.zeros_like(x)
        assert lib().emul_dcnotch(self.B, self.M, x.shape[2], _vp(x), _vp(y), _vp(self.mem), ctypes.c_float(self.radius)) == 0
        return y

    def firbank(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.zeros_like(x)
        mean = np.zeros(x.shape[:2], dtype=np.float32)
        assert lib().emul_firbank(self.B, self.M, x.shape[1], self.L, _vp(x), _vp(y), _vp(mean), _vp(self.coef),
                                  _vp(self.cache[self.cur]), _vp(self.cache[self.cur ^ 1])) == 0
        self.cur ^= 1
        return y, mean


class EmulTdFilter:
    """sample-wise NLMS / RLS block program (ds_tdfilter.hpp)."""

    def __init__(self, mode, L, mu, lam=0.9998, norm=1, batch=1):
        self.mode, self.L, self.mu, self.lam, self.norm, self.B = mode, L, mu, lam, norm, batch
        self.w = np.zeros((batch, L), np.float32)
        self.buf = np.zeros((batch, L), np.float32)
        self.P = np.tile(np.eye(L, dtype=np.float32) * 1000.0, (batch, 1, 1)) if mode == 1 else np.zeros((1,), np.float32)

    def update(self, x, d, p=1.0):
        x = np.ascontiguousarray(x, np.float32); d = np.ascontiguousarray(d, np.float32)
        err = np.zeros_like(x)
        f = ctypes.c_float
        assert lib().emul_tdfilter(self.mode, self.B, x.shape[1], self.L, _vp(x), _vp(d), _vp(err), _vp(self.w), _vp(self.buf),
                                   _vp(self.P), f(self.mu), f(1e-4), f(p), f(self.lam), self.norm) == 0
        return err


class EmulFdaf:
    """overlap-save FDAF block program (ds_fdaf.hpp); kind 0 plain / 1 clamped blocking filter / 2 norm-limited canceller."""

    def __init__(self, filter_len, n_channels=1, mu=0.01, alpha=0.9, kind=0, constrain=True, non_causal=False,
                 weight_norm=False, batch=1):
        self.L, self.C, self.mu, self.alpha, self.kind, self.B = filter_len, n_channels, mu, alpha, kind, batch
        self.constrain, self.non_causal, self.weight_norm = int(constrain), int(non_causal), int(weight_norm)
        K, C, L = filter_len + 1, n_channels, filter_len
        self.K = K
        self.two_path = False
        self.state = np.zeros((batch, 2 * C * K + K + C * L + L // 2 + 2 * C * K), np.float32)     # ... | foreground [C][K] (two_path)

    @property
    def W(self):
        C, K = self.C, self.K
        return self.state[:, :2 * C * K].copy().view(np.complex64).reshape(self.B, C, K)

    @property
    def P(self):
        C, K = self.C, self.K
        return self.state[:, 2 * C * K:2 * C * K + K]

    def update(self, x, d, p=None, fir_truncate=None, want_w=True):
        """x [B, T*L, C], d [B, T*L], p None | [B, T] | [B, T, K] -> (err [B, T*L], w [B, L, C])."""
        x = np.ascontiguousarray(x, np.float32); d = np.ascontiguousarray(d, np.float32)
        T = x.shape[1] // self.L
        err = np.zeros_like(d)
        w = np.zeros((self.B, self.L, self.C), np.float32)
        pm = 0 if p is None else (1 if np.ndim(p) == 2 else 2)
        pp = np.zeros(1, np.float32) if p is None else np.ascontiguousarray(p, np.float32)
        f = ctypes.c_float
        lib().emul_set_two_path(int(self.two_path))
        rc = lib().emul_fdaf(2 * self.L, self.B, T, self.C, self.kind, self.constrain, self.non_causal, self.weight_norm,
                             -1 if fir_truncate is None else int(fir_truncate), pm, f(self.mu), f(self.alpha), _vp(x), _vp(d), _vp(pp),
                             _vp(err), _vp(w) if want_w else None, _vp(self.state))
        lib().emul_set_two_path(0)
        assert rc == 0
        return err, w


class EmulAdaptiveFrames:
    """adaptivebeamfomer's frame loop as a frame-level operator (op_adaptive in ds_ops.hpp)."""

    def __init__(self, nfft, M, steer, batch=1, L=15, method=2):
        self.B, self.K, self.M, self.L, self.method = batch, nfft // 2 + 1, M, L, method
        self.KP = (self.K + 7) & ~7                       # ds_core.hpp plane_len()
        self.NF = M * M + 5
        self.st = np.zeros((batch, self.NF, self.KP), dtype=np.float32)
        self.steer = np.ascontiguousarray(steer, dtype=np.complex64)
        self.frm, self.ell = 0, 1

    def run(self, Z, gain=None):
        Z = np.ascontiguousarray(Z, dtype=np.complex64)
        T = Z.shape[1]
        Y = np.zeros((self.B, T, self.K), dtype=np.complex64)
        g = None if gain is None else np.ascontiguousarray(gain, dtype=np.float32)
        f = ctypes.c_float
        rc = lib().emul_adaptive_frames(self.B, self.K, T, self.M, _vp(self.st), self.NF, _vp(Z), _vp(g), _vp(Y), _vp(self.steer),
                                        self.frm, self.ell, self.L, self.method, f(0.9998), f(0.4), f(1e-6))
        assert rc == 0
        for _ in range(T):
            if self.frm != 0 and self.ell % self.L == 0:
                self.ell = 0
            self.frm += 1
            self.ell += 1
        return Y


class EmulWpe:
    """lane-parallel RLS-WPE block program (ds_wpe.hpp); state = one block per (utterance, bin), see wpe_bin_floats()."""

    def __init__(self, nfft, C, N, batch=1, lam=0.998):
        self.B, self.K, self.C, self.N, self.lam = batch, nfft // 2 + 1, C, N, lam
        CN = C * N
        from distantspeech_amd.ops import wpe_block_layout
        self.layout = wpe_block_layout(C, N)
        self.SB = self.layout["floats"]
        self.state = np.zeros((batch, self.K, self.SB), np.float32)
        for i in range(CN):
            self.state[:, :, 2 * (i * (i + 1) // 2 + i)] = 1e-3          # diagonal of the packed upper triangle

    def run(self, xd, d):
        xd = np.ascontiguousarray(xd, np.complex64); d = np.ascontiguousarray(d, np.complex64)
        err = np.zeros_like(d)
        assert lib().emul_wpe(self.B, self.K, d.shape[1], self.C, self.N, _vp(xd), _vp(d), _vp(err), _vp(self.state),
                              ctypes.c_float(self.lam)) == 0
        return err
