/* mvdr_stream.c — the C-ABI of libdsenh.so from plain C (no Python, no HIP headers): one adaptive-MVDR handle fed hop by hop the way
 * the reference's realtime shell feeds adaptivebeamfomer.process (DistantSpeech/realtime/realtime_processing.py:78-84).
 *
 *   gcc -O2 -I include examples/c/mvdr_stream.c -o mvdr_stream -L distantspeech_amd -ldsenh -lm -Wl,-rpath,$PWD/distantspeech_amd
 *   ./mvdr_stream x.f32 y.f32 n_samples        (x: float32 [4][n_samples] channel-major, y: float32 [n_samples])
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dsenh.h"

#define M 4
#define NFFT 512
#define HOP 256
#define K (NFFT / 2 + 1)

int main(int argc, char** argv) {
    if (argc != 4) { fprintf(stderr, "usage: %s x.f32 y.f32 n_samples\n", argv[0]); return 2; }
    const int n = atoi(argv[3]);
    if (n <= 0 || n % HOP) { fprintf(stderr, "n_samples must be a positive multiple of %d\n", HOP); return 2; }
    float* x = (float*)malloc(sizeof(float) * M * (size_t)n);
    float* y = (float*)malloc(sizeof(float) * (size_t)n);
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(x, sizeof(float), (size_t)M * n, f) != (size_t)M * n) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
    fclose(f);

    ds_config cfg;
    memset(&cfg, 0, sizeof cfg);                    /* zeros = the reference's defaults */
    cfg.struct_size = (int32_t)sizeof cfg;
    cfg.algo = DS_ALGO_ADAPTIVE; cfg.n_mics = M; cfg.nfft = NFFT; cfg.hop = HOP; cfg.batch = 1; cfg.device = -1;
    ds_handle* h = NULL;
    int rc = ds_create(&cfg, &h);
    if (rc) { fprintf(stderr, "ds_create: %s (%s)\n", ds_strerror(rc), ds_last_error(NULL)); return 1; }

    /* steering vector of the 4-microphone circular array (r = 0.032 m) towards 197 degrees: a[k][m] = exp(-j w_k tao_m),
     * adaptivebeamformer.py:52,84 */
    const double PI = 3.14159265358979323846, az = 197.0 / 180.0 * PI;
    float steer[K * M * 2];
    for (int k = 0; k < K; ++k)
        for (int m = 0; m < M; ++m) {
            const double tao = -0.032 * cos(0.0) * cos(az - m * PI / 2.0) / 343.0;
            const double w = 2.0 * PI * k * 16000.0 / NFFT;
            steer[2 * (k * M + m)] = (float)cos(-w * tao);
            steer[2 * (k * M + m) + 1] = (float)sin(-w * tao);
        }
    rc = ds_set_steering(h, steer, 0);
    if (!rc) rc = ds_set_param_i(h, DS_PARAM_METHOD, DS_METHOD_MVDR);
    if (rc) { fprintf(stderr, "set-up: %s\n", ds_last_error(h)); return 1; }

    float hop_in[M * HOP];
    for (int s = 0; s < n && !rc; s += HOP) {       /* one hop per call: the streaming-callback contract */
        for (int m = 0; m < M; ++m) memcpy(hop_in + m * HOP, x + (size_t)m * n + s, sizeof(float) * HOP);
        rc = ds_process(h, hop_in, DS_LAYOUT_CHANNELS_SAMPLES, HOP, y + s);
    }
    if (rc) { fprintf(stderr, "ds_process: %s\n", ds_last_error(h)); return 1; }
    f = fopen(argv[2], "wb");
    fwrite(y, sizeof(float), (size_t)n, f);
    fclose(f);
    double acc = 0.0;
    for (int i = 0; i < n; ++i) acc += (double)y[i] * y[i];
    printf("processed %d samples in %d calls, output rms %.6f\n", n, n / HOP, sqrt(acc / n));
    ds_destroy(h);
    free(x); free(y);
    return 0;
}
