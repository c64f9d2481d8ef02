/* A C99 caller of the C-ABI (include/rt_analyze.h): what a foreign binding does, without Python in between.
 *   abi_smoke <iq.bin> <n_streams> <n_samples> <sample_rate> <nperseg> <window.bin>
 * reads n_streams * n_samples complex64 from the first file and nperseg float32 window coefficients followed by the
 * float32 PSD scale 1 / (fs * sum w^2) from the last one (what SciPy forms for complex64 input,
 * scipy/signal/_spectral_py.py:2083-2087), analyses one buffer per stream with the reference's default parameters
 * (-90 dBW, 5 dB, 8..40 ms) and prints one line per record:
 *   stream fi start end shadowed max_p mean_p std_db row_mean            (floats as hex bit patterns)
 * exit code 3 = no GPU (rt_create says RT_E_NO_DEVICE), 0 = ok.  Built and driven by tests/test_c_caller.py. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rt_analyze.h"

static unsigned bits(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    return u;
}

int main(int argc, char **argv) {
    if (argc != 7) return 64;
    const int n_streams = atoi(argv[2]);
    const long n_samples = atol(argv[3]);
    const double fs = atof(argv[4]);
    const int nperseg = atoi(argv[5]);
    if (rt_abi_version() != RT_ABI_VERSION) {
        fprintf(stderr, "ABI %d, header %d\n", rt_abi_version(), RT_ABI_VERSION);
        return 65;
    }
    float *w = (float *)malloc(sizeof(float) * ((size_t)nperseg + 1));
    FILE *fw = fopen(argv[6], "rb");
    if (!fw || fread(w, 4, (size_t)nperseg + 1, fw) != (size_t)nperseg + 1) return 72;
    fclose(fw);
    rt_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.device = 0;
    cfg.n_streams = n_streams;
    cfg.nperseg = nperseg;
    cfg.mode = RT_MODE_AUTO;
    cfg.max_samples = n_samples;
    cfg.sample_rate = fs;
    cfg.window = w;
    cfg.scale = w[nperseg];
    cfg.threshold = (float)pow(10.0, -90.0 / 10.0);
    cfg.snr_threshold = (float)pow(10.0, 5.0 / 10.0);
    cfg.calibration_db = 0.f;
    cfg.min_duration_s = 8.0 / 1000;
    cfg.max_duration_s = 40.0 / 1000;
    rt_handle *h = NULL;
    int rc = rt_create(&cfg, &h);
    if (rc == RT_E_NO_DEVICE) {
        fprintf(stderr, "no GPU: %s\n", rt_last_error(NULL));
        return 3;
    }
    if (rc != RT_OK) {
        fprintf(stderr, "rt_create: %d %s\n", rc, rt_last_error(NULL));
        return 66;
    }
    const size_t n = (size_t)n_streams * (size_t)n_samples;
    float *iq = (float *)malloc(n * 8);
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(iq, 8, n, f) != n) return 67;
    fclose(f);
    rc = rt_process_host(h, iq, n_samples, n_samples);
    if (rc != RT_OK) {
        fprintf(stderr, "rt_process_host: %d %s\n", rc, rt_last_error(h));
        return 68;
    }
    size_t count = 0;
    rc = rt_fetch(h, NULL, 0, &count); /* size query */
    if (rc != RT_OK) return 69;
    rt_record *rec = (rt_record *)malloc(sizeof(rt_record) * (count ? count : 1));
    if (count) {
        rc = rt_fetch(h, rec, count, &count);
        if (rc != RT_OK) return 70;
    }
    rt_call_info info;
    if (rt_get_call_info(h, &info) != RT_OK) return 71;
    for (size_t i = 0; i < count; ++i)
        printf("%d %d %d %d %d %08x %08x %08x %08x\n", rec[i].stream, rec[i].fi, rec[i].start, rec[i].end, rec[i].shadowed, bits(rec[i].max_p),
               bits(rec[i].mean_p), bits(rec[i].std_db), bits(rec[i].row_mean));
    fprintf(stderr, "%zu records, %d segments, mode %d\n", count, info.n_seg, info.mode_used);
    rt_destroy(h);
    free(rec);
    free(iq);
    free(w);
    return 0;
}
