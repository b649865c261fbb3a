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
#include <type_traits>
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

// The inverse of a Hermitian positive-definite A kept as the INVERSE OF ITS CHOLESKY FACTOR (round 6; the double twin of ds_core.hpp's
// Chol::invert / apply / trace_with): A = L L^H, Li = L^-1 (lower triangular, real diagonal), A^-1 = Li^H Li.  Everything the notebook operator
// takes from inv(Phi_vv + dv I) is a product with it — tr(A^-1 Phi_yy) = sum over the rows r_i of Li of the quadratic forms r_i Phi_yy r_i^H,
// A^-1 y = Li^H (Li y), a^H A^-1 a = |Li a|^2 — so the explicit inverse (2 M^2 doubles, both triangles) never has to exist beside the factor:
// M^2 doubles live instead of 3 M^2 at the peak of herm_inverse_d, for about the same arithmetic.
template <int M> struct CholInvD {
    double id[M];                               // diagonal of Li = 1 / diagonal of L
    cd lo[M * (M - 1) / 2 + 1];                 // strictly lower triangle, row-wise: (i, j), i > j, at i (i - 1) / 2 + j
    DS_HD cd& at(int i, int j) { return lo[i * (i - 1) / 2 + j]; }
    DS_HD const cd& at(int i, int j) const { return lo[i * (i - 1) / 2 + j]; }
    // a(i, j), i >= j: the lower triangle of A (a(j, j).x its real diagonal)
    template <class Get> DS_HD void factor_invert(Get a) {
#pragma unroll
        for (int j = 0; j < M; ++j) {
            double s = a(j, j).x;
#pragma unroll
            for (int q = 0; q < j; ++q) s = fmad_(-at(j, q).x, at(j, q).x, fmad_(-at(j, q).y, at(j, q).y, s));
            s = s > 1e-300 ? s : 1e-300;
            const double r = rsqrt_fast_d(s);           // (the hardware's seed + two Newton steps: a couple of ulp; IEEE sqrt and division are 23 instructions a pivot)
            id[j] = r;
#pragma unroll
            for (int i = j + 1; i < M; ++i) {
                cd t = a(i, j);
#pragma unroll
                for (int q = 0; q < j; ++q) t = cdfnmac(t, at(i, q), at(j, q));
                at(i, j) = cdscale(t, r);
            }
        }
        // L -> L^-1 in place, column by column: entry (i, j) from row i of L to the right of column j (not overwritten yet) and the finished
        // part of column j
#pragma unroll
        for (int j = 0; j < M; ++j)
#pragma unroll
            for (int i = j + 1; i < M; ++i) {
                cd t = cdscale(at(i, j), id[j]);
#pragma unroll
                for (int q = j + 1; q < i; ++q) t = cdfma(t, at(i, q), at(q, j));
                at(i, j) = cdscale(t, -id[i]);
            }
    }
    // u = Li b
    DS_HD void lower(const cd* b, cd* u) const {
#pragma unroll
        for (int i = 0; i < M; ++i) {
            cd t = cdscale(b[i], id[i]);
#pragma unroll
            for (int j = 0; j < i; ++j) t = cdfma(t, at(i, j), b[j]);
            u[i] = t;
        }
    }
    // v = Li^H u
    DS_HD void upper(const cd* u, cd* v) const {
#pragma unroll
        for (int j = 0; j < M; ++j) {
            cd t = cdscale(u[j], id[j]);
#pragma unroll
            for (int i = j + 1; i < M; ++i) t = cdfmac(t, u[i], at(i, j));          // + conj(Li_ij) u_i
            v[j] = t;
        }
    }
    // Re tr(A^-1 R), R Hermitian through its getter R(j, k) (any order of the indices)
    template <class Get> DS_HD double trace_with(Get R) const {
        double tr = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            double q = id[i] * id[i] * R(i, i).x, c2 = 0.0;
#pragma unroll
            for (int j = 0; j < i; ++j) {
                const cd rj = at(i, j);
                q = fmad_(cdabs2(rj), R(j, j).x, q);
                { const cd rji = R(j, i); c2 = fmad_(rj.x * id[i], rji.x, fmad_(-(rj.y * id[i]), rji.y, c2)); }   // Re(r_ij R_ji r_ii), r_ii real
#pragma unroll
                for (int k = j + 1; k < i; ++k) {
                    const cd c = cdmulc(rj, at(i, k)), rjk = R(j, k);               // r_ij conj(r_ik)
                    c2 = fmad_(c.x, rjk.x, fmad_(-c.y, rjk.y, c2));                 // Re(r_ij R_jk conj(r_ik))
                }
            }
            tr += fmad_(2.0, c2, q);
        }
        return tr;
    }
    // entry (i, j) of A^-1 = sum over q >= max(i, j) of conj(Li_qi) Li_qj   (only where a caller wants the matrix itself)
    DS_HD cd inverse_entry(int i, int j) const {
        cd t = mkd(0.0, 0.0);
#pragma unroll
        for (int q = 0; q < M; ++q) {
            if (q < i || q < j) continue;
            const cd a = q == i ? mkd(id[q], 0.0) : at(q, i), b = q == j ? mkd(id[q], 0.0) : at(q, j);
            t = cdfmac(t, b, a);                                                    // + b conj(a)
        }
        return t;
    }
};

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

// The same vector by a DIRECT method (round 6): Householder tridiagonalisation -> the largest root of the tridiagonal's characteristic
// polynomial by Laguerre's iteration from the Gershgorin bound -> inverse iteration on the tridiagonal -> the reflectors applied back.
// steering() keeps ONE eigenvector (beamformer.py:24: eigh(...)[1][:, :, -1]); the Jacobi solve above diagonalises the whole matrix for it —
// 5-7 sweeps of M (M - 1) / 2 rotations, each on A and on the M x M accumulator V: 7 345 vector instructions and 428 registers in the
// 6-microphone notebook operator, 87 % of its step.  Here, for M = 6: four reflectors (rank-2 updates of shrinking trailing blocks), five or
// six Laguerre steps on three-term recurrences of length M, two triangular solves of a tridiagonal, four reflector applications — about a
// fifth of the arithmetic, no accumulator matrix.  Why each piece:
//  * Phi_xx = Phi_yy - Phi_vv is Hermitian and INDEFINITE; the eigenvalue wanted is the algebraically largest.  A real-rooted polynomial
//    with positive leading coefficient has p, p', p'' > 0 to the right of its largest root, and Laguerre's iteration started there decreases
//    monotonically onto that root (cubically at a simple root, linearly at a multiple one) without ever passing it: no bracketing, no
//    Sturm counts, and no way to land on another eigenvalue.
//  * With the shift at (a hair above) the largest eigenvalue, shift I - T is positive semi-definite: its L D L^H factorisation needs no
//    pivoting, the pivots are the ratios p_k / p_{k-1} of the recurrence; the one that vanishes is floored at rounding level (a perturbation of
//    that size of T) and two solves turn any start vector into the eigenvector to eps ||T|| / gap — the conditioning any method has, LAPACK's
//    included.  A tridiagonal that falls apart into blocks needs no special case: the block that owns the eigenvalue is amplified by 1 / eps.
//  * The matrix is scaled to unit size first (covariances of quiet recordings are 1e-12), entries below 1e-17 of it are dropped — the
//    threshold the Jacobi solve stops at.  A matrix that is diagonal to that threshold (Phi_xx = 0 exactly during McSpp's first ten
//    frames, mcspp.py:273-275) takes the Jacobi solve's exit: the unit vector of the largest diagonal entry, the LAST one on ties, like
//    eigh's ascending order.
// Same result as herm_principal_d up to the conditioning of the eigenvector (tests/test_kernel_emul.py holds the two against each other and
// against numpy on random, indefinite, clustered and block-diagonal matrices).  Only the upper triangle and the diagonal of A are read.
// (the matrix comes through a getter `a(i, j)`, i <= j, called twice per entry — once for the scale, once for the scaled copy — so that a caller
// whose matrix is a difference of fp32 state words never holds an unscaled double copy of it beside the working triangle)
// `between()` runs between the two passes over the getter: a caller passes the fence that keeps the compiler from merging the two reads of an
// entry into one held double (which is the unscaled copy this arrangement is there to avoid)
// `mag2(i, j)` (optional): |a_ij|^2 in whatever precision the caller has it cheaply (the notebook operator: the fp32 state words) — it only sizes
// the matrix (any scale within a few per cent normalises it as well) and decides the diagonal exit, where an entry that is exactly zero is
// exactly zero in any precision
struct NoFence { DS_HD void operator()() const {} };
struct NoMag { };
template <int M, class Get, class Fence = NoFence, class Mag = NoMag>
DS_HD void herm_principal_direct_get_d(Get a, cd* v, int* laguerre_steps = nullptr, Fence between = Fence(), Mag mag2 = Mag()) {
    // ---- scale, and the diagonal exit -----------------------------------------------------------------------------------------------------
    double dmax = 0.0, offmax = 0.0;
#pragma unroll
    for (int i = 0; i < M; ++i) {
        double t;
        if constexpr (std::is_same<Mag, NoMag>::value) t = fabs(a(i, i).x); else t = sqrt((double)mag2(i, i));
        dmax = t > dmax ? t : dmax;
    }
#pragma unroll
    for (int i = 0; i < M - 1; ++i)
#pragma unroll
        for (int j = i + 1; j < M; ++j) {
            double m2;
            if constexpr (std::is_same<Mag, NoMag>::value) m2 = cdabs2(a(i, j)); else m2 = (double)mag2(i, j);
            offmax = m2 > offmax ? m2 : offmax;
        }
    if (!(offmax > 1e-34 * dmax * dmax && offmax > 1e-300)) {
        int best = 0;
        double wmax = a(0, 0).x;
#pragma unroll
        for (int i = 1; i < M; ++i) { const double t = a(i, i).x; if (t >= wmax) { wmax = t; best = i; } }
#pragma unroll
        for (int k = 0; k < M; ++k) v[k] = mkd(k == best ? 1.0 : 0.0, 0.0);
        return;
    }
    const double big2 = dmax * dmax > offmax ? dmax * dmax : offmax;
    const double sc = rsqrt_fast_d(big2);                                         // 1 / max |a_ij|
    between();
    // lower triangle of the scaled matrix: L[i][j], i >= j  (a_ij = conj(a_ji))
    cd Lw[M][M];
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) { const cd t = a(j, i); Lw[i][j] = i == j ? mkd(t.x * sc, 0.0) : mkd(t.x * sc, -(t.y * sc)); }
    // ---- Householder: H_k = I - beta_k u_k u_k^H on rows k + 1 .., u_k kept in column k below the diagonal ----------------------------------
    double d[M], e2[M], beta[M];              // diagonal, |subdiagonal|^2 (e2[k]: between k and k + 1), 2 / (u^H u)
    cd e[M];                                  // subdiagonal T[k + 1][k]
#pragma unroll
    for (int k = 0; k < M - 2; ++k) {
        constexpr double TINY2 = 1e-34;       // (1e-17)^2 of the unit-size matrix
        double s2 = 0.0;
#pragma unroll
        for (int i = k + 2; i < M; ++i) s2 += cdabs2(Lw[i][k]);
        const cd x0 = Lw[k + 1][k];
        const double a02 = cdabs2(x0);
        d[k] = Lw[k][k].x;
        if (!(s2 > TINY2)) {                  // nothing below the subdiagonal: no reflector
            beta[k] = 0.0; e[k] = x0; e2[k] = a02;
            continue;
        }
        const double n2 = s2 + a02, rn = rsqrt_fast_d(n2), nx = n2 * rn;          // ||x||
        const double ra0 = a02 > 1e-300 ? rsqrt_fast_d(a02) : 0.0, a0 = a02 * ra0;
        const cd ph = a02 > 1e-300 ? cdscale(x0, ra0) : mkd(1.0, 0.0);            // e^{j arg x0}
        e[k] = cdscale(ph, -nx); e2[k] = n2;                                     // H x = -e^{j arg x0} ||x|| e_0
        Lw[k + 1][k] = cdscale(ph, a0 + nx);                                      // u_0 = x_0 + e^{j arg x0} ||x||
        beta[k] = rcp_fast_d(nx * (nx + a0));                                     // 2 / (u^H u)
        // trailing block B = A[k+1.., k+1..]: p = beta B u, K = beta (u^H p) / 2, q = p - K u, B -= u q^H + q u^H
        cd pv[M];
        double uhp = 0.0;
#pragma unroll
        for (int i = k + 1; i < M; ++i) {
            cd acc = mkd(0.0, 0.0);
#pragma unroll
            for (int j = k + 1; j < M; ++j) acc = j <= i ? cdfma(acc, Lw[i][j], Lw[j][k]) : cdfma(acc, cdconj(Lw[j][i]), Lw[j][k]);
            pv[i] = cdscale(acc, beta[k]);
            uhp = fmad_(Lw[i][k].x, pv[i].x, fmad_(Lw[i][k].y, pv[i].y, uhp));    // Re(conj(u_i) p_i); the sum is real for a Hermitian B
        }
        const double Kh = 0.5 * beta[k] * uhp;
#pragma unroll
        for (int i = k + 1; i < M; ++i) pv[i] = mkd(fmad_(-Kh, Lw[i][k].x, pv[i].x), fmad_(-Kh, Lw[i][k].y, pv[i].y));   // q
#pragma unroll
        for (int i = k + 1; i < M; ++i)
#pragma unroll
            for (int j = k + 1; j <= i; ++j) {
                cd t = cdfnmac(Lw[i][j], Lw[i][k], pv[j]);                        // - u_i conj(q_j)
                t = cdfnmac(t, pv[i], Lw[j][k]);                                  // - q_i conj(u_j)
                if (i == j) t.y = 0.0;
                Lw[i][j] = t;
            }
    }
    d[M - 2] = Lw[M - 2][M - 2].x;
    d[M - 1] = Lw[M - 1][M - 1].x;
    e[M - 2] = Lw[M - 1][M - 2]; e2[M - 2] = cdabs2(e[M - 2]);
    // ---- largest eigenvalue of the tridiagonal: Laguerre from the Gershgorin bound ---------------------------------------------------------
    double ae[M];
#pragma unroll
    for (int k = 0; k < M - 1; ++k) ae[k] = e2[k] > 1e-300 ? e2[k] * rsqrt_fast_d(e2[k]) : 0.0;
    double gersh = d[0] + ae[0];
#pragma unroll
    for (int k = 1; k < M; ++k) { const double g = d[k] + ae[k - 1] + (k < M - 1 ? ae[k] : 0.0); gersh = g > gersh ? g : gersh; }
    gersh += 4e-15;
    double lam = gersh;
    {   // ... or the Wolkowicz-Styan bound mean + sqrt(n - 1) * deviation of the eigenvalues (from the traces of T and T^2), whichever is lower: it
        // is EXACT for a rank-one matrix, which is what Phi_xx is close to whenever the source is active.  The variance is a difference of two
        // numbers of unit size: what its cancellation can have lost is added back, so that the bound stays a bound
        double tr = 0.0, tr2 = 0.0;
#pragma unroll
        for (int k = 0; k < M; ++k) { tr += d[k]; tr2 = fmad_(d[k], d[k], tr2); }
#pragma unroll
        for (int k = 0; k < M - 1; ++k) tr2 = fmad_(2.0, e2[k], tr2);
        const double mean = tr * (1.0 / M), ms = tr2 * (1.0 / M);
        double var = fmad_(-mean, mean, ms);
        var = (var > 0.0 ? var : 0.0) + 8e-16 * ms;
        const double ws = fmad_(sqrt((double)(M - 1) * var), 1.0 + 1e-14, mean) + 4e-15;
        lam = ws < lam ? ws : lam;
    }
    // to the right of the largest root every leading minor p_k of lam I - T is positive (that IS lam I - T positive definite): checked at
    // every iterate for free, the recurrence produces them.  A start that fails it (never seen; the bounds are bounds) restarts from Gershgorin's
#pragma unroll 1
    for (int it = 0; it < 12; ++it) {
        double p0 = 1.0, p1 = lam - d[0], q0 = 0.0, q1 = 1.0, r0 = 0.0, r1 = 0.0;    // p, p', p'' of orders k - 1 and k
        bool pd = p1 > 0.0;
#pragma unroll
        for (int k = 1; k < M; ++k) {
            const double x = lam - d[k], b = e2[k - 1];
            const double p2 = fmad_(x, p1, -(b * p0));
            const double q2 = fmad_(x, q1, fmad_(-b, q0, p1));
            const double r2 = fmad_(x, r1, fmad_(-b, r0, 2.0 * q1));
            p0 = p1; p1 = p2; q0 = q1; q1 = q2; r0 = r1; r1 = r2;
            pd = pd && p1 > 0.0;
        }
        if (!pd) {
            if (it == 0 && lam < gersh) { lam = gersh; continue; }
            break;                                                                 // on the root to rounding
        }
        const double ip = rcp_fast_d(p1), G = q1 * ip, Hh = fmad_(G, G, -(r1 * ip));
        double disc = (double)(M - 1) * fmad_((double)M, Hh, -(G * G));
        disc = disc > 0.0 ? disc : 0.0;
        const double sq = disc > 1e-300 ? disc * rsqrt_fast_d(disc) : 0.0;
        const double step = (double)M * rcp_fast_d(G + sq);
        lam -= step;
        if (laguerre_steps) *laguerre_steps = it + 1;
        if (!(step > 3e-16)) break;                                                 // (of a unit-size matrix)
    }
    // ---- inverse iteration: (lam I - T) = L D L^H, pivots floored at rounding level ---------------------------------------------------------
    double idl[M];                            // 1 / pivot
    cd l[M];                                  // L[k][k - 1]
    {
        // (the computed root may sit a few 1e-14 BELOW the eigenvalue — the recurrence's rounding over p' — and the last pivot is then a small
        // negative number that carries the eigenvector: only a pivot of rounding size is replaced, its sign is kept otherwise)
        auto inv_pivot = [](double piv) { return fabs(piv) > 3e-16 ? rcp_fast_d(fabs(piv)) * (piv < 0.0 ? -1.0 : 1.0) : 1.0 / 3e-16; };
        idl[0] = inv_pivot(lam - d[0]);
#pragma unroll
        for (int k = 1; k < M; ++k) {
            l[k] = cdscale(e[k - 1], -idl[k - 1]);
            idl[k] = inv_pivot(fmad_(-e2[k - 1], idl[k - 1], lam - d[k]));
        }
    }
    cd x[M];
#pragma unroll
    for (int k = 0; k < M; ++k) x[k] = mkd(1.0, 0.0);                             // first pass: L y = b chosen so that y = ones
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
        if (pass > 0) {
#pragma unroll
            for (int k = 1; k < M; ++k) x[k] = cdfnma(x[k], l[k], x[k - 1]);      // forward: y_k = b_k - l_k y_{k-1}
        }
#pragma unroll
        for (int k = 0; k < M; ++k) x[k] = cdscale(x[k], idl[k]);
#pragma unroll
        for (int k = M - 2; k >= 0; --k) x[k] = cdfnma(x[k], cdconj(l[k + 1]), x[k + 1]);   // backward: x_k = z_k - conj(l_{k+1}) x_{k+1}
        double m2 = 0.0;
#pragma unroll
        for (int k = 0; k < M; ++k) m2 += cdabs2(x[k]);
        const double rs = m2 > 1e-300 ? rsqrt_fast_d(m2) : 1.0;
#pragma unroll
        for (int k = 0; k < M; ++k) x[k] = cdscale(x[k], rs);
    }
    // ---- back through the reflectors: v = H_0 H_1 ... H_{M-3} x --------------------------------------------------------------------------
#pragma unroll
    for (int k = M - 3; k >= 0; --k) {
        cd uhx = mkd(0.0, 0.0);
#pragma unroll
        for (int i = k + 1; i < M; ++i) uhx = cdfmac(uhx, x[i], Lw[i][k]);         // u^H x
        uhx = cdscale(uhx, beta[k]);
#pragma unroll
        for (int i = k + 1; i < M; ++i) x[i] = cdfnma(x[i], Lw[i][k], uhx);
    }
    double m2 = 0.0;
#pragma unroll
    for (int k = 0; k < M; ++k) m2 += cdabs2(x[k]);
    const double rs = m2 > 1e-300 ? 1.0 / sqrt(m2) : 1.0;
    // unit norm, exp(-j angle(v0)); a first component at rounding level (an eigenvector that lives in a trailing block) is the Jacobi solve's exact
    // zero, whose angle is 0 (beamformer.py:27-28)
    const double n0 = sqrt(cdabs2(x[0]));
    const bool v0_zero = !(n0 * rs > 1e-14);
    if (v0_zero) x[0] = mkd(0.0, 0.0);
    const cd ph = v0_zero ? mkd(rs, 0.0) : cdscale(cdconj(x[0]), rs / n0);
#pragma unroll
    for (int k = 0; k < M; ++k) v[k] = cdmul(x[k], ph);
}

template <int M> DS_HD void herm_principal_direct_d(const cd (&A)[M][M], cd* v, int* laguerre_steps = nullptr) {
    herm_principal_direct_get_d<M>([&](int i, int j) { return A[i][j]; }, v, laguerre_steps);
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
