// ds_linalg64.hpp — double-precision small complex-Hermitian linear algebra in registers (M <= 8).
//
// The notebook online-MVDR flow (example/mvdr.ipynb cell 4: McSpp.estimation -> steering(Phi_yy - Phi_vv) ->
// compute_mvdr_weight) is conditioning-limited: Phi_xx = Phi_yy - Phi_vv cancels to a small remainder whose principal
// eigenvector steers the beamformer, and inv(Phi_vv + dv I) is taken of matrices that are nearly rank one at the low bins of
// real recordings.  The reference does all of it in complex128 (beamformer.py:10-31, mcspp.py:201-242).  With the carried state
// in fp32 and this arithmetic in fp64 the enhanced signal is within 4e-5 RMS of the reference on the golden recordings (fp32
// arithmetic: 8e-4); MI355X's fp64 vector rate makes that cheap next to the operator's HBM traffic.
//
// Same contraction discipline as ds_core.hpp: every fused multiply-add is explicit.
#pragma once
#include "ds_core.hpp"

namespace ds {

struct cd { double x, y; };

DS_HD double fmad_(double a, double b, double c) { return __builtin_fma(a, b, c); }
DS_HD double dmin_(double a, double b) { return a < b ? a : b; }
DS_HD double dmax_(double a, double b) { return a > b ? a : b; }
DS_HD cd mkd(double a, double b) { cd r; r.x = a; r.y = b; return r; }
DS_HD cd to_cd(cf a) { return mkd((double)a.x, (double)a.y); }
DS_HD cd cdsub(cd a, cd b) { return mkd(a.x - b.x, a.y - b.y); }
DS_HD cd cdconj(cd a) { return mkd(a.x, -a.y); }
DS_HD cd cdscale(cd a, double s) { return mkd(a.x * s, a.y * s); }
// 1 / sqrt(x) and 1 / x for the Jacobi rotation angles (x a normal, positive number): the hardware's seed (v_rsq_f64 / v_rcp_f64) and two
// Newton steps — a couple of ulp, where IEEE sqrt and division cost 14 instructions each (scaling, class tests, fix-up) and a rotation
// needed six of them.  An angle that is off by an ulp leaves an off-diagonal residue of that size for the next sweep; c^2 + s^2 = 1 holds to
// the same couple of ulp.  On the host the plain expressions.
DS_HD double rsqrt_fast_d(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    y = fmad_(y, fmad_(-(h * y), y, 0.5), y);
    y = fmad_(y, fmad_(-(h * y), y, 0.5), y);
    return y;
#else
    return 1.0 / sqrt(x);
#endif
}
DS_HD double rcp_fast_d(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rcp(x);
    y = fmad_(y, fmad_(-x, y, 1.0), y);
    y = fmad_(y, fmad_(-x, y, 1.0), y);
    return y;
#else
    return 1.0 / x;
#endif
}
DS_HD double cdabs2(cd a) { return fmad_(a.x, a.x, a.y * a.y); }
DS_HD cd cdmul(cd a, cd b) { return mkd(fmad_(a.x, b.x, -(a.y * b.y)), fmad_(a.x, b.y, a.y * b.x)); }
DS_HD cd cdmulc(cd a, cd b) { return mkd(fmad_(a.x, b.x, a.y * b.y), fmad_(a.y, b.x, -(a.x * b.y))); }    // a * conj(b)
DS_HD cd cdfma(cd acc, cd a, cd b) {         // acc + a * b
    return mkd(fmad_(a.x, b.x, fmad_(-a.y, b.y, acc.x)), fmad_(a.x, b.y, fmad_(a.y, b.x, acc.y)));
}
DS_HD cd cdfmac(cd acc, cd a, cd b) {        // acc + a * conj(b)
    return mkd(fmad_(a.x, b.x, fmad_(a.y, b.y, acc.x)), fmad_(a.y, b.x, fmad_(-a.x, b.y, acc.y)));
}
DS_HD cd cdfnma(cd acc, cd a, cd b) {        // acc - a * b
    return mkd(fmad_(-a.x, b.x, fmad_(a.y, b.y, acc.x)), fmad_(-a.x, b.y, fmad_(-a.y, b.x, acc.y)));
}
DS_HD cd cdfnmac(cd acc, cd a, cd b) {       // acc - a * conj(b)
    return mkd(fmad_(-a.x, b.x, fmad_(-a.y, b.y, acc.x)), fmad_(-a.y, b.x, fmad_(a.x, b.y, acc.y)));
}
DS_HD cd cddiv(cd a, cd b) {
    const double d = 1.0 / cdabs2(b);
    const cd n = cdmulc(a, b);
    return mkd(n.x * d, n.y * d);
}

// Hermitian-packed fp32 state (diag reals d[M], strictly-upper complex o[]) -> full double matrix
template <int M> DS_HD void herm_unpack_d(const float* d, const float* o, cd (&A)[M][M]) {
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = 0; j < M; ++j) A[i][j] = to_cd(herm_get<M>(d, o, i, j));
}

// inverse of the Hermitian positive-definite A: Cholesky A = L L^H, Linv, inv = Linv^H Linv  (np.linalg.inv, mcspp.py:214,224-226)
template <int M> DS_HD void herm_inverse_d(const cd (&A)[M][M], cd (&inv)[M][M]) {
    double invd[M];
    cd L[M][M], Li[M][M];
#pragma unroll
    for (int j = 0; j < M; ++j) {
        double s = A[j][j].x;
#pragma unroll
        for (int q = 0; q < j; ++q) s = fmad_(-L[j][q].x, L[j][q].x, fmad_(-L[j][q].y, L[j][q].y, s));
        s = s > 1e-300 ? s : 1e-300;
        const double r = 1.0 / sqrt(s);
        invd[j] = r;
#pragma unroll
        for (int i = j + 1; i < M; ++i) {
            cd a = A[i][j];
#pragma unroll
            for (int q = 0; q < j; ++q) a = cdfnmac(a, L[i][q], L[j][q]);
            L[i][j] = cdscale(a, r);
        }
    }
#pragma unroll
    for (int c = 0; c < M; ++c)
#pragma unroll
        for (int i = c; i < M; ++i) {
            cd t = (i == c) ? mkd(1.0, 0.0) : mkd(0.0, 0.0);
#pragma unroll
            for (int q = c; q < i; ++q) t = cdfnma(t, L[i][q], Li[q][c]);
            Li[i][c] = cdscale(t, invd[i]);
        }
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = i; j < M; ++j) {
            cd t = mkd(0.0, 0.0);
#pragma unroll
            for (int q = j; q < M; ++q) t = cdfmac(t, Li[q][j], Li[q][i]);    // sum conj(Li_qi) Li_qj
            inv[i][j] = t;
            inv[j][i] = cdconj(t);
        }
}

// principal eigenvector (largest eigenvalue) of the Hermitian A (destroyed) by cyclic complex Jacobi, phase-normalised by
// element 0 (beamformer/beamformer.py:10-31: np.linalg.eigh(...)[1][:, :, -1] / exp(j angle(v0))).  Sweeps stop when no rotation
// of a sweep was larger than rounding (quadratic convergence: 5-7 sweeps in double for M <= 8).
// Round 5 (this solve is 150 k of the notebook-MVDR operator's 151 k vector instructions per frame at 6 microphones):
//  * only the UPPER triangle of A is kept and updated (a_ij, i < j; the lower one was its mirror image, written after every rotation);
//  * a rotation whose off-diagonal entry is already below the stopping threshold is skipped — its angle is < 1e-17, it changes nothing in
//    double; before, the sweep that FOUND the matrix converged still carried out all its M (M - 1) / 2 rotations on such entries.
template <int M> DS_HD void herm_principal_d(cd (&A)[M][M], cd* v) {
    cd V[M][M];
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = 0; j < M; ++j) V[i][j] = mkd(i == j ? 1.0 : 0.0, 0.0);
    // upper(i, j) for any i != j: the stored entry or the conjugate of its mirror
    auto up = [&](int i, int j) -> cd { return i < j ? A[i][j] : cdconj(A[j][i]); };
    for (int sweep = 0; sweep < 16; ++sweep) {
        double offmax = 0.0, dmax = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i) { const double a = fabs(A[i][i].x); dmax = a > dmax ? a : dmax; }
        const double tiny = 1e-34 * dmax * dmax;                                      // |a_pq| <= 1e-17 max|a_ii|
#pragma unroll
        for (int p = 0; p < M - 1; ++p)
#pragma unroll
            for (int q = p + 1; q < M; ++q) {
                const cd apq = A[p][q];
                const double mag2 = cdabs2(apq);
                offmax = mag2 > offmax ? mag2 : offmax;
                if (mag2 > tiny && mag2 > 1e-300) {
                    // t = sgn(tau) / (|tau| + sqrt(tau^2 + 1)), tau = (a_qq - a_pp) / (2 |a_pq|), written in w = t / |a_pq| = sgn(d) / (|d| +
                    // sqrt(d^2 + |a_pq|^2)), d = (a_qq - a_pp) / 2: one square root, one reciprocal and one reciprocal square root per rotation
                    // instead of three square roots and three divisions (|a_pq| itself is never needed: s e^{j phi} = c w a_pq, t |a_pq| = w |a_pq|^2)
                    const double app = A[p][p].x, aqq = A[q][q].x;
                    const double del = 0.5 * (aqq - app);
                    const double r2 = fmad_(del, del, mag2);
                    const double w = (del >= 0.0 ? 1.0 : -1.0) * rcp_fast_d(fabs(del) + r2 * rsqrt_fast_d(r2));
                    const double c = rsqrt_fast_d(fmad_(w * w, mag2, 1.0));
                    const double t = w, mag = mag2;                                   // (t mag below = w |a_pq|^2)
                    const cd se = cdscale(apq, c * w);                                // s e^{j phi}
#pragma unroll
                    for (int k = 0; k < M; ++k) {
                        if (k != p && k != q) {
                            const cd akp = up(k, p), akq = up(k, q);
                            const cd np_ = cdfnmac(cdscale(akp, c), akq, se);        // c akp - s conj(e) akq
                            const cd nq_ = cdfma(cdscale(akq, c), se, akp);          // s e akp + c akq
                            if (k < p) A[k][p] = np_; else A[p][k] = cdconj(np_);
                            if (k < q) A[k][q] = nq_; else A[q][k] = cdconj(nq_);
                        }
                    }
                    A[p][p] = mkd(fmad_(-t, mag, app), 0.0);
                    A[q][q] = mkd(fmad_(t, mag, aqq), 0.0);
                    A[p][q] = mkd(0.0, 0.0);
#pragma unroll
                    for (int k = 0; k < M; ++k) {
                        const cd vkp = V[k][p], vkq = V[k][q];
                        V[k][p] = cdfnmac(cdscale(vkp, c), vkq, se);
                        V[k][q] = cdfma(cdscale(vkq, c), se, vkp);
                    }
                }
            }
        if (offmax <= tiny) break;
    }
    int best = 0;
    double wmax = A[0][0].x;
#pragma unroll
    for (int i = 1; i < M; ++i) if (A[i][i].x >= wmax) { wmax = A[i][i].x; best = i; }   // ties: the last one, like eigh's ascending order
    cd v0 = mkd(1.0, 0.0);
#pragma unroll
    for (int i = 0; i < M; ++i) if (i == best) v0 = V[0][i];
    const double n0 = sqrt(cdabs2(v0));
    const cd ph = n0 > 0.0 ? cdscale(cdconj(v0), 1.0 / n0) : mkd(1.0, 0.0);         // exp(-j angle(v0))
#pragma unroll
    for (int k = 0; k < M; ++k) {
        cd vk = mkd(0.0, 0.0);
#pragma unroll
        for (int i = 0; i < M; ++i) if (i == best) vk = V[k][i];
        v[k] = cdmul(vk, ph);
    }
}

// w = R^-1 a / (a^H R^-1 a)   (beamformer/beamformer.py:133-155)
template <int M> DS_HD void mvdr_weight_d(const cd (&Rinv)[M][M], const cd* a, cd* w) {
    cd num[M];
    cd den = mkd(0.0, 0.0);
#pragma unroll
    for (int i = 0; i < M; ++i) {
        cd t = mkd(0.0, 0.0);
#pragma unroll
        for (int j = 0; j < M; ++j) t = cdfma(t, Rinv[i][j], a[j]);
        num[i] = t;
        den = cdfmac(den, t, a[i]);                                                 // + conj(a_i) num_i
    }
#pragma unroll
    for (int i = 0; i < M; ++i) w[i] = cddiv(num[i], den);
}

// Generalised Hermitian eigenproblem A v = lambda B v, B positive definite: the eigenvector of the LARGEST eigenvalue, normalised to
// v^H B v = 1 like scipy.linalg.eigh(a, b)[1][:, -1] (beamformer/beamformer.py:79-97, get_gev_vector).  Cholesky whitening B = L L^H,
// C = L^-1 A L^-H, Jacobi on C (herm_principal_d: unit-norm y, phase fixed by y_0 real and positive), v = L^-H y.  The PHASE of an
// eigenvector is LAPACK's business in the reference (the caller's phase_correction removes it from bin to bin and leaves the first
// bin's): here the whitened vector's first component is real and positive, and the tests compare up to that phase.
// Returns false where B is not positive definite (the reference's LinAlgError branch: ones / trace(B) * M).
template <int M> DS_HD bool herm_gev_principal_d(const cd (&A)[M][M], const cd (&Bm)[M][M], cd* v) {
    double invd[M];
    cd L[M][M], Li[M][M];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < M; ++j) {
        double s = Bm[j][j].x;
#pragma unroll
        for (int q = 0; q < j; ++q) s = fmad_(-L[j][q].x, L[j][q].x, fmad_(-L[j][q].y, L[j][q].y, s));
        if (!(s > 0.0)) { ok = false; s = 1e-300; }
        const double r = 1.0 / sqrt(s);
        invd[j] = r;
#pragma unroll
        for (int i = j + 1; i < M; ++i) {
            cd a = Bm[i][j];
#pragma unroll
            for (int q = 0; q < j; ++q) a = cdfnmac(a, L[i][q], L[j][q]);
            L[i][j] = cdscale(a, r);
        }
    }
    if (!ok) return false;
#pragma unroll
    for (int c = 0; c < M; ++c)                                   // Li = L^-1 (lower)
#pragma unroll
        for (int i = 0; i < M; ++i) {
            if (i < c) { Li[i][c] = mkd(0.0, 0.0); continue; }
            cd t = (i == c) ? mkd(1.0, 0.0) : mkd(0.0, 0.0);
#pragma unroll
            for (int q = c; q < i; ++q) t = cdfnma(t, L[i][q], Li[q][c]);
            Li[i][c] = cdscale(t, invd[i]);
        }
    cd T[M][M], C[M][M];
#pragma unroll
    for (int i = 0; i < M; ++i)                                   // T = Li A
#pragma unroll
        for (int j = 0; j < M; ++j) {
            cd t = mkd(0.0, 0.0);
#pragma unroll
            for (int q = 0; q <= i; ++q) t = cdfma(t, Li[i][q], A[q][j]);
            T[i][j] = t;
        }
#pragma unroll
    for (int i = 0; i < M; ++i)                                   // C = T Li^H (Hermitian: the upper triangle mirrors the lower one)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            cd t = mkd(0.0, 0.0);
#pragma unroll
            for (int q = 0; q <= j; ++q) t = cdfmac(t, T[i][q], Li[j][q]);
            if (i == j) t.y = 0.0;
            C[i][j] = t;
            C[j][i] = cdconj(t);
        }
    cd y[M];
    herm_principal_d<M>(C, y);
#pragma unroll
    for (int i = 0; i < M; ++i) {                                 // v = Li^H y
        cd t = mkd(0.0, 0.0);
#pragma unroll
        for (int q = i; q < M; ++q) t = cdfma(t, cdconj(Li[q][i]), y[q]);
        v[i] = t;
    }
    return true;
}

}  // namespace ds
