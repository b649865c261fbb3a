/* ds_oracle_mvdr.c — plain-C (double precision) restatement of the reference's adaptive-MVDR frame loop.
 *
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.  Used only by tests/ (checked against the reference's golden vectors
 * G4) and by bench.py's cpu_baseline leg (timed on the host cores).  The product (libdsenh.so) never links it.
 *
 * Follows, statement by statement (paths relative to /root/reference/DistantSpeech):
 *   Transform.stft / istft                transform/transform.py:407-481  (complex64 rounding :212, float32 OLA :359)
 *   NoiseEstimationMCRA.estimation        noise_estimation/mcra.py:27-77, NoiseEstimationBase.py:56-60
 *   adaptivebeamfomer.process (method 2)  beamformer/adaptivebeamformer.py:44-128
 *   beamformer.getweights('MVDR')         beamformer/beamformer.py:325-326
 * numpy.fft.rfft/irfft -> an in-file radix-2 complex FFT (double); numpy.linalg.inv -> Gauss-Jordan with partial pivoting.
 * Parity pin: tests/test_oracle_c.py holds this file to the reference's golden vectors (tests/golden/g4_*.npz).
 */
#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MAXM 8
typedef double complex cplx;

typedef struct {
    int M, nfft, hop, K;
    double *win, *prev_in /*[M][hop]*/, *prev_out /*[hop]*/;
    cplx* tw;                     /* twiddles exp(-2 pi j k / nfft) */
    /* MCRA (mcra.py) */
    double *S, *Smin, *Stmp, *p, *lambda_d;
    int ell, frm_cnt, L;
    /* adaptive beamformer state */
    cplx *Rvv /*[K][M][M]*/, *Rvv_inv, *a /*[K][M]*/;
    /* scratch */
    cplx *X /*[M][K]*/, *buf, *Y;
    double* frame;
} ds_oracle;

static void fft_inplace(cplx* a, int n, const cplx* tw, int twn, int inverse) {
    for (int i = 1, j = 0; i < n; ++i) {                     /* bit reversal */
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { cplx t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    for (int len = 2; len <= n; len <<= 1) {
        const int step = twn / len;
        for (int i = 0; i < n; i += len)
            for (int k = 0; k < len / 2; ++k) {
                cplx w = tw[k * step];
                if (inverse) w = conj(w);
                const cplx u = a[i + k], v = a[i + k + len / 2] * w;
                a[i + k] = u + v;
                a[i + k + len / 2] = u - v;
            }
    }
}

ds_oracle* dso_create(int M, int nfft, int hop, const double* steer_re_im /* [K][M][2] */, int mcra_L) {
    ds_oracle* o = (ds_oracle*)calloc(1, sizeof(ds_oracle));
    const int K = nfft / 2 + 1;
    o->M = M; o->nfft = nfft; o->hop = hop; o->K = K; o->L = mcra_L; o->ell = 1;   /* NoiseEstimationBase.py:18 */
    o->win = (double*)malloc(sizeof(double) * nfft);
    for (int n = 0; n < nfft; ++n) o->win[n] = sqrt(0.5 - 0.5 * cos(2.0 * M_PI * n / nfft));   /* transform.py:418-419 */
    o->tw = (cplx*)malloc(sizeof(cplx) * nfft);
    for (int k = 0; k < nfft; ++k) o->tw[k] = cexp(-2.0 * M_PI * I * k / nfft);
    o->prev_in = (double*)calloc((size_t)M * hop, sizeof(double));
    o->prev_out = (double*)calloc(hop, sizeof(double));
    o->S = (double*)calloc(K, sizeof(double)); o->Smin = (double*)calloc(K, sizeof(double));
    o->Stmp = (double*)calloc(K, sizeof(double)); o->p = (double*)calloc(K, sizeof(double));
    o->lambda_d = (double*)calloc(K, sizeof(double));
    o->Rvv = (cplx*)calloc((size_t)K * M * M, sizeof(cplx));
    o->Rvv_inv = (cplx*)calloc((size_t)K * M * M, sizeof(cplx));
    o->a = (cplx*)malloc(sizeof(cplx) * K * M);
    for (int i = 0; i < K * M; ++i) o->a[i] = steer_re_im[2 * i] + I * steer_re_im[2 * i + 1];
    o->X = (cplx*)malloc(sizeof(cplx) * M * K);
    o->buf = (cplx*)malloc(sizeof(cplx) * nfft);
    o->Y = (cplx*)malloc(sizeof(cplx) * K);
    o->frame = (double*)malloc(sizeof(double) * nfft);
    return o;
}

void dso_destroy(ds_oracle* o) {
    if (!o) return;
    free(o->win); free(o->tw); free(o->prev_in); free(o->prev_out); free(o->S); free(o->Smin); free(o->Stmp); free(o->p);
    free(o->lambda_d); free(o->Rvv); free(o->Rvv_inv); free(o->a); free(o->X); free(o->buf); free(o->Y); free(o->frame);
    free(o);
}

static void mcra(ds_oracle* o, const double* Y) {                 /* mcra.py:27-77 */
    const int K = o->K;
    for (int k = 0; k < K - 1; ++k) {
        if (o->frm_cnt == 0) {
            o->Smin[k] = Y[k]; o->Stmp[k] = Y[k]; o->lambda_d[k] = Y[k];
        } else {
            if (k == 0) { o->p[0] = 0; continue; }
            const double Sf = Y[k - 1] * 0.25 + Y[k] * 0.5 + Y[k + 1] * 0.25;
            o->S[k] = 0.8 * o->S[k] + (1 - 0.8) * Sf;
            o->Smin[k] = fmin(o->Smin[k], o->S[k]);
            o->Stmp[k] = fmin(o->Stmp[k], o->S[k]);
            if (o->ell % o->L == 0) { o->Smin[k] = fmin(o->Stmp[k], o->S[k]); o->Stmp[k] = o->S[k]; o->ell = 0; }
            const double Sr = o->S[k] / (o->Smin[k] + 1e-6);
            const double Ind = Sr > 5 ? 1.0 : 0.0;
            o->p[k] = 0.2 * o->p[k] + (1 - 0.2) * Ind;
        }
        if (o->frm_cnt < o->L * 2) o->p[k] = 0.0;
    }
    for (int k = 0; k < K; ++k) o->p[k] = fmax(fmin(o->p[k], 0.999), 1e-3);
    o->frm_cnt += 1;
    o->lambda_d[K - 1] = 1e-8;
    o->ell += 1;
    for (int k = 0; k < K; ++k) {                                /* NoiseEstimationBase.py:56-60 */
        const double at = 0.95 + (1 - 0.95) * o->p[k];
        o->lambda_d[k] = at * o->lambda_d[k] + (1 - at) * Y[k];
    }
}

static void invert(const cplx* A, cplx* inv, int M) {           /* numpy.linalg.inv: Gauss-Jordan, partial pivoting */
    cplx w[MAXM][2 * MAXM];
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < M; ++j) { w[i][j] = A[i * M + j]; w[i][M + j] = (i == j) ? 1.0 : 0.0; }
    for (int c = 0; c < M; ++c) {
        int piv = c;
        for (int r = c + 1; r < M; ++r) if (cabs(w[r][c]) > cabs(w[piv][c])) piv = r;
        if (piv != c) for (int j = 0; j < 2 * M; ++j) { cplx t = w[c][j]; w[c][j] = w[piv][j]; w[piv][j] = t; }
        const cplx d = 1.0 / w[c][c];
        for (int j = 0; j < 2 * M; ++j) w[c][j] *= d;
        for (int r = 0; r < M; ++r) if (r != c) {
            const cplx f = w[r][c];
            for (int j = 0; j < 2 * M; ++j) w[r][j] -= f * w[c][j];
        }
    }
    for (int i = 0; i < M; ++i) for (int j = 0; j < M; ++j) inv[i * M + j] = w[i][M + j];
}

/* one hop: x [M][hop] -> y [hop]   (adaptivebeamformer.py:44-128 called with one hop, method = 2) */
void dso_process_hop(ds_oracle* o, const float* x, float* y) {
    const int M = o->M, N = o->nfft, hop = o->hop, K = o->K, ov = N - hop;
    for (int m = 0; m < M; ++m) {                                /* Transform.stft :430-453 */
        for (int n = 0; n < ov; ++n) o->buf[n] = o->win[n] * o->prev_in[m * ov + n];
        for (int n = 0; n < hop; ++n) o->buf[ov + n] = o->win[ov + n] * (double)x[m * hop + n];
        for (int n = 0; n < hop; ++n) o->prev_in[m * ov + n] = (double)x[m * hop + n];   /* ov == hop */
        fft_inplace(o->buf, N, o->tw, N, 0);
        for (int k = 0; k < K; ++k)                              /* complex64 storage :212 */
            o->X[m * K + k] = (double)(float)creal(o->buf[k]) + I * (double)(float)cimag(o->buf[k]);
    }
    for (int k = 0; k < K; ++k) o->frame[k] = cabs(o->X[k] * conj(o->X[k]));            /* :81 */
    mcra(o, o->frame);
    for (int k = 0; k < K; ++k) {
        cplx z[MAXM];
        for (int m = 0; m < M; ++m) z[m] = o->X[m * K + k];
        cplx* R = o->Rvv + (size_t)k * M * M;
        cplx* Ri = o->Rvv_inv + (size_t)k * M * M;
        if (o->p[k] < 0.4) {                                     /* :94-104 */
            cplx Rl[MAXM * MAXM];
            for (int i = 0; i < M; ++i)
                for (int j = 0; j < M; ++j) {
                    R[i * M + j] = 0.9998 * R[i * M + j] + (1 - 0.9998) * (z[i] * conj(z[j]));
                    Rl[i * M + j] = R[i * M + j] + ((i == j) ? 1e-6 : 0.0);
                }
            invert(Rl, Ri, M);
        }
        cplx num[MAXM], den = 0;                                 /* beamformer.py:325-326 */
        for (int i = 0; i < M; ++i) {
            num[i] = 0;
            for (int j = 0; j < M; ++j) num[i] += Ri[i * M + j] * o->a[k * M + j];
            den += conj(o->a[k * M + i]) * num[i];
        }
        cplx acc = 0;
        for (int m = 0; m < M; ++m) acc += conj(num[m] / den) * z[m];                   /* :119-120 */
        o->Y[k] = acc;
    }
    /* Transform.istft :455-481 (one frame): irfft ignores Im Y[0], Im Y[N/2] */
    o->buf[0] = creal(o->Y[0]);
    for (int k = 1; k < K - 1; ++k) { o->buf[k] = o->Y[k]; o->buf[N - k] = conj(o->Y[k]); }
    o->buf[N / 2] = creal(o->Y[K - 1]);
    fft_inplace(o->buf, N, o->tw, N, 1);
    double W0 = 0;
    for (int n = 0; n < N; ++n) W0 += o->win[n] * o->win[n];
    for (int n = 0; n < N; ++n) {
        const float f = (float)(o->win[n] * creal(o->buf[n]) / N);                      /* float32 OLA buffer :359 */
        if (n < ov) {
            const float s = (float)((double)f + o->prev_out[n]);                         /* :476 */
            y[n] = (float)((double)s * hop / W0);                                        /* :479 */
        } else {
            o->prev_out[n - ov] = (double)f;                                             /* :477 */
        }
    }
}

/* T hops of one utterance: x [M][T*hop] -> y [T*hop] */
void dso_process(ds_oracle* o, const float* x, int n_samples, float* y) {
    const int T = n_samples / o->hop, M = o->M, hop = o->hop;
    float* xh = (float*)malloc(sizeof(float) * M * hop);
    for (int t = 0; t < T; ++t) {
        for (int m = 0; m < M; ++m) memcpy(xh + m * hop, x + (size_t)m * n_samples + (size_t)t * hop, sizeof(float) * hop);
        dso_process_hop(o, xh, y + (size_t)t * hop);
    }
    free(xh);
}

void dso_get_p(const ds_oracle* o, double* p) { memcpy(p, o->p, sizeof(double) * o->K); }
