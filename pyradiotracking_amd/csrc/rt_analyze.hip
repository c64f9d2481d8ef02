// rt_analyze.hip -- host side of the C-ABI declared in include/rt_analyze.h:
// handle, device scratch, kernel launches.  gfx950 only; fails loudly when no
// GPU is usable (there is no CPU fallback in the product path).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "rt_kernels.h"

using namespace rt;

namespace {

thread_local std::string g_create_error;

struct CallCtx {
    bool valid = false;
    const void *iq = nullptr;
    int64_t n_samples = 0;
    int64_t stream_stride = 0;
    int n_seg = 0;
    int tail_read = 0;     // index of the tail buffer holding the previous buffer's columns
    int n_seg_last = -1;   // columns of the previous buffer (-1: none)
    int mode_used = 0;
    bool fell_back = false;
    bool is_extract = false;
};

}  // namespace

struct rt_handle {
    rt_config cfg{};
    int R3 = 1, N = 256, LG = 16, GPW = 16;
    int K = 1;             // tail columns
    int stride = 1;        // probe stride
    int L = 32;            // segments per chunk
    int max_seg = 0;       // T for max_samples
    int max_chunks = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;

    float *d_window = nullptr;
    cf *d_tw1 = nullptr, *d_tw2 = nullptr;
    float *d_psum = nullptr;
    float *d_tail[2] = {nullptr, nullptr};
    uint2 *d_hot = nullptr;
    uint32_t *d_hot_count = nullptr;
    rt_record *d_records = nullptr;
    int64_t pool_cap = 0;
    int32_t *d_rec_offset = nullptr, *d_rec_count = nullptr;
    unsigned long long *d_counters = nullptr;  // 4 words
    float *d_spec = nullptr;                   // lazily allocated dense spectrogram
    void *d_iq_stage = nullptr;                // for rt_process_host
    size_t iq_stage_bytes = 0;

    // pinned host mirrors
    unsigned long long *h_counters = nullptr;
    int32_t *h_rec_offset = nullptr, *h_rec_count = nullptr;
    rt_record *h_records = nullptr;
    size_t h_records_cap = 0;

    int hot_cap = 8192, rec_cap = 1024;
    size_t lds_sparse = 0, lds_dense = 0;

    int tail_cur = 0;      // tail buffer holding the most recent completed buffer
    int n_seg_last = -1;

    CallCtx call;
    rt_call_info info{};
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool timing = false;
};

namespace {

#define RT_HIP(h, expr)                                                                       \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            (h)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                     \
            return RT_E_HIP;                                                                  \
        }                                                                                     \
    } while (0)

int fail_create(int code, const std::string &msg) {
    g_create_error = msg;
    return code;
}

size_t rec_lds_bytes(int rec_cap) { return (size_t)rec_cap * (8 + 8 + sizeof(rt_record)) + 16; }

int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

template <int MODE>
void launch_stft(rt_handle *h, const StftParams &p, int blocks) {
    switch (h->R3) {
        case 1: hipLaunchKernelGGL((stft_scan<1, MODE>), dim3(blocks), dim3(kBlock), 0, h->stream, p); break;
        case 2: hipLaunchKernelGGL((stft_scan<2, MODE>), dim3(blocks), dim3(kBlock), 0, h->stream, p); break;
        case 4: hipLaunchKernelGGL((stft_scan<4, MODE>), dim3(blocks), dim3(kBlock), 0, h->stream, p); break;
        case 8: hipLaunchKernelGGL((stft_scan<8, MODE>), dim3(blocks), dim3(kBlock), 0, h->stream, p); break;
        default: hipLaunchKernelGGL((stft_scan<16, MODE>), dim3(blocks), dim3(kBlock), 0, h->stream, p); break;
    }
}

int choose_chunk(const rt_handle *h, int n_seg) {
    if (h->cfg.segs_per_chunk > 0) return h->cfg.segs_per_chunk;
    // enough workgroups to fill 256 CUs several times over, halo overhead <= 1/L
    int L = 32;
    while (L > 4) {
        const int64_t chunks = (n_seg + L - 1) / L;
        const int64_t blocks = (int64_t)h->cfg.n_streams * ((chunks + h->GPW - 1) / h->GPW);
        if (blocks >= 2048) break;
        L >>= 1;
    }
    return L;
}

StftParams make_stft_params(rt_handle *h, const void *iq, int64_t stream_stride, int n_seg, int tail_write) {
    StftParams p{};
    p.iq = static_cast<const cf *>(iq);
    p.stream_stride = stream_stride;
    p.n_streams = h->cfg.n_streams;
    p.n_seg = n_seg;
    p.segs_per_chunk = h->L;
    p.chunks = (n_seg + h->L - 1) / h->L;
    p.blocks_per_stream = (p.chunks + h->GPW - 1) / h->GPW;
    p.tail_cols = h->K;
    p.window = h->d_window;
    p.tw1 = h->d_tw1;
    p.tw2 = h->d_tw2;
    p.scale = h->cfg.scale;
    p.thr = h->cfg.threshold;
    p.psum = h->d_psum;
    p.tail = h->d_tail[tail_write];
    p.spec = h->d_spec;
    p.hot = h->d_hot;
    p.hot_count = h->d_hot_count;
    p.hot_cap = h->hot_cap;
    return p;
}

DetectArgs make_detect_args(rt_handle *h, int n_seg, int n_bins, int n_seg_last) {
    DetectArgs a{};
    a.dp.n_seg = n_seg;
    a.dp.n_seg_last = n_seg_last;
    a.dp.tail_cols = h->K;
    a.dp.stride = h->stride;
    a.dp.nperseg = h->N;
    a.dp.thr = h->cfg.threshold;
    a.dp.snr = h->cfg.snr_threshold;
    a.dp.cal_db = h->cfg.calibration_db;
    a.dp.fs = h->cfg.sample_rate;
    a.dp.min_d = h->cfg.min_duration_s;
    a.dp.max_d = h->cfg.max_duration_s;
    a.n_streams = h->cfg.n_streams;
    a.n_bins = n_bins;
    a.hot = h->d_hot;
    a.hot_count = h->d_hot_count;
    a.hot_cap = h->hot_cap;
    a.psum = h->d_psum;
    a.records = h->d_records;
    a.pool_cap = h->pool_cap;
    a.rec_cap = h->rec_cap;
    a.rec_offset = h->d_rec_offset;
    a.rec_count = h->d_rec_count;
    a.counters = h->d_counters;
    return a;
}

int ensure_dense_spec(rt_handle *h) {
    if (h->d_spec) return RT_OK;
    const size_t bytes = (size_t)h->cfg.n_streams * (size_t)h->max_seg * (size_t)h->N * sizeof(float);
    hipError_t e = hipMalloc(&h->d_spec, bytes ? bytes : 4);
    if (e != hipSuccess) {
        h->err = "dense spectrogram scratch (" + std::to_string(bytes) + " bytes): " + hipGetErrorString(e);
        h->d_spec = nullptr;
        return RT_E_NOMEM;
    }
    return RT_OK;
}

int enqueue_readback(rt_handle *h) {
    RT_HIP(h, hipMemcpyAsync(h->h_counters, h->d_counters, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                             h->stream));
    const size_t sb = (size_t)h->cfg.n_streams * sizeof(int32_t);
    RT_HIP(h, hipMemcpyAsync(h->h_rec_offset, h->d_rec_offset, sb, hipMemcpyDeviceToHost, h->stream));
    RT_HIP(h, hipMemcpyAsync(h->h_rec_count, h->d_rec_count, sb, hipMemcpyDeviceToHost, h->stream));
    return RT_OK;
}

// enqueue scan + detect for the call described by h->call
int enqueue_analysis(rt_handle *h, bool dense) {
    const CallCtx &c = h->call;
    const int tail_write = 1 - c.tail_read;
    StftParams sp = make_stft_params(h, c.iq, c.stream_stride, c.n_seg, tail_write);
    if (sp.chunks > h->max_chunks) {
        h->err = "internal: chunk count exceeds scratch";
        return RT_E_INVALID;
    }
    const int blocks = h->cfg.n_streams * sp.blocks_per_stream;
    RT_HIP(h, hipMemsetAsync(h->d_counters, 0, 4 * sizeof(unsigned long long), h->stream));
    if (!dense) RT_HIP(h, hipMemsetAsync(h->d_hot_count, 0, (size_t)h->cfg.n_streams * sizeof(uint32_t), h->stream));
    if (dense) {
        int rc = ensure_dense_spec(h);
        if (rc != RT_OK) return rc;
        sp.spec = h->d_spec;
    }
    if (h->timing) RT_HIP(h, hipEventRecord(h->ev[0], h->stream));
    if (dense)
        launch_stft<1>(h, sp, blocks);
    else
        launch_stft<0>(h, sp, blocks);
    RT_HIP(h, hipGetLastError());
    if (h->timing) RT_HIP(h, hipEventRecord(h->ev[1], h->stream));

    DetectArgs a = make_detect_args(h, c.n_seg, h->N, c.n_seg_last);
    a.prev = h->d_tail[c.tail_read];
    a.prev_cols = h->K;
    a.chunks = sp.blocks_per_stream;
    a.spec = h->d_spec;
    if (dense)
        hipLaunchKernelGGL(detect_dense, dim3(h->cfg.n_streams), dim3(kDetBlock), h->lds_dense, h->stream, a);
    else
        hipLaunchKernelGGL(detect_sparse, dim3(h->cfg.n_streams), dim3(kDetBlock), h->lds_sparse, h->stream, a);
    RT_HIP(h, hipGetLastError());
    if (h->timing) RT_HIP(h, hipEventRecord(h->ev[2], h->stream));
    return enqueue_readback(h);
}

}  // namespace

extern "C" {

int rt_abi_version(void) { return RT_ABI_VERSION; }

int rt_device_count(int *count) {
    if (!count) return RT_E_INVALID;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        g_create_error = std::string("hipGetDeviceCount: ") + hipGetErrorString(e);
        return RT_E_NO_DEVICE;
    }
    *count = n;
    return RT_OK;
}

const char *rt_last_error(rt_handle *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

void rt_destroy(rt_handle *h) {
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    (void)hipFree(h->d_window);
    (void)hipFree(h->d_tw1);
    (void)hipFree(h->d_tw2);
    (void)hipFree(h->d_psum);
    (void)hipFree(h->d_tail[0]);
    (void)hipFree(h->d_tail[1]);
    (void)hipFree(h->d_hot);
    (void)hipFree(h->d_hot_count);
    (void)hipFree(h->d_records);
    (void)hipFree(h->d_rec_offset);
    (void)hipFree(h->d_rec_count);
    (void)hipFree(h->d_counters);
    (void)hipFree(h->d_spec);
    (void)hipFree(h->d_iq_stage);
    (void)hipHostFree(h->h_counters);
    (void)hipHostFree(h->h_rec_offset);
    (void)hipHostFree(h->h_rec_count);
    (void)hipHostFree(h->h_records);
    for (auto &e : h->ev)
        if (e) (void)hipEventDestroy(e);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int rt_create(const rt_config *cfg, rt_handle **out) {
    if (!cfg || !out) return fail_create(RT_E_INVALID, "null argument");
    *out = nullptr;
    if (cfg->n_streams < 1 || cfg->max_samples < 0 || !cfg->window || !(cfg->sample_rate > 0))
        return fail_create(RT_E_INVALID, "n_streams, max_samples, window and sample_rate must be set");
    if (!(cfg->max_duration_s >= 0) || !(cfg->min_duration_s >= 0))
        return fail_create(RT_E_INVALID, "durations must be non-negative");
    int R3 = 0;
    for (int r : {1, 2, 4, 8, 16})
        if (cfg->nperseg == 256 * r) R3 = r;
    if (!R3) return fail_create(RT_E_UNSUPPORTED, "nperseg must be one of 256, 512, 1024, 2048, 4096");
    if (cfg->mode < RT_MODE_AUTO || cfg->mode > RT_MODE_SPARSE) return fail_create(RT_E_INVALID, "bad mode");

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1)
        return fail_create(RT_E_NO_DEVICE, std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "count 0"));
    if (cfg->device < 0 || cfg->device >= ndev) return fail_create(RT_E_INVALID, "device ordinal out of range");
    e = hipSetDevice(cfg->device);
    if (e != hipSuccess) return fail_create(RT_E_NO_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));

    rt_handle *h = new (std::nothrow) rt_handle();
    if (!h) return fail_create(RT_E_NOMEM, "out of host memory");
    h->cfg = *cfg;
    h->cfg.window = nullptr;
    h->R3 = R3;
    h->N = cfg->nperseg;
    h->LG = 16 * R3;
    h->GPW = kBlock / h->LG;
    h->timing = (cfg->flags & RT_FLAG_TIMING) != 0;
    h->hot_cap = cfg->hot_capacity > 0 ? cfg->hot_capacity : 8192;
    h->rec_cap = cfg->record_capacity > 0 ? cfg->record_capacity : 1024;
    h->stride = probe_stride(h->N, cfg->sample_rate, cfg->min_duration_s);
    {
        const double hop = seg_time(1, h->N, cfg->sample_rate) - seg_time(0, h->N, cfg->sample_rate);
        const double k = std::floor(cfg->max_duration_s / hop) + 2.0;
        h->K = (int)std::min(k, 1.0e6);
        if (h->K < 1) h->K = 1;
    }
    h->max_seg = (int)(cfg->max_samples / h->N);
    h->L = choose_chunk(h, h->max_seg);  // fixed per handle so the scratch bound holds for every call
    h->max_chunks = std::max(1, (h->max_seg + h->L - 1) / h->L);
    const int max_blocks_per_stream = (h->max_chunks + h->GPW - 1) / h->GPW;
    if ((int64_t)h->max_seg * h->N > 0xFFFFFFFFll) {
        delete h;
        return fail_create(RT_E_UNSUPPORTED, "max_samples too large for 32-bit cell keys");
    }
    h->lds_dense = rec_lds_bytes(h->rec_cap);
    h->lds_sparse = rec_lds_bytes(h->rec_cap) + sizeof(float) * ((h->N + 3) & ~3) + (size_t)next_pow2(h->hot_cap) * 9;
    if (h->lds_sparse > 160 * 1024 || h->lds_dense > 160 * 1024) {
        delete h;
        return fail_create(RT_E_INVALID, "hot_capacity/record_capacity do not fit the 160 KiB LDS of a CU");
    }

    auto fail = [&](int code, const std::string &msg) {
        std::string m = msg;
        rt_destroy(h);
        return fail_create(code, m);
    };
#define RT_CREATE_HIP(expr)                                                                  \
    do {                                                                                     \
        hipError_t e2_ = (expr);                                                             \
        if (e2_ != hipSuccess) return fail(RT_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e2_)); \
    } while (0)

    if (cfg->hip_stream) {
        h->stream = static_cast<hipStream_t>(cfg->hip_stream);
    } else {
        RT_CREATE_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        h->own_stream = true;
    }

    const int S = cfg->n_streams, N = h->N, LG = h->LG;
    // window and twiddle tables (twiddles in double, rounded once to float32)
    std::vector<cf> tw1((size_t)LG * 16), tw2((size_t)R3 * 16);
    const double two_pi = 6.283185307179586476925286766559;
    for (int a = 0; a < LG; ++a)
        for (int k1 = 0; k1 < 16; ++k1) {
            const double ang = -two_pi * (double)((a * k1) % N) / (double)N;
            tw1[(size_t)a * 16 + k1] = cf{(float)std::cos(ang), (float)std::sin(ang)};
        }
    for (int b = 0; b < R3; ++b)
        for (int q1 = 0; q1 < 16; ++q1) {
            const double ang = -two_pi * (double)((b * q1) % LG) / (double)LG;
            tw2[(size_t)b * 16 + q1] = cf{(float)std::cos(ang), (float)std::sin(ang)};
        }
    RT_CREATE_HIP(hipMalloc(&h->d_window, sizeof(float) * N));
    RT_CREATE_HIP(hipMalloc(&h->d_tw1, sizeof(cf) * tw1.size()));
    RT_CREATE_HIP(hipMalloc(&h->d_tw2, sizeof(cf) * tw2.size()));
    RT_CREATE_HIP(hipMemcpy(h->d_window, cfg->window, sizeof(float) * N, hipMemcpyHostToDevice));
    RT_CREATE_HIP(hipMemcpy(h->d_tw1, tw1.data(), sizeof(cf) * tw1.size(), hipMemcpyHostToDevice));
    RT_CREATE_HIP(hipMemcpy(h->d_tw2, tw2.data(), sizeof(cf) * tw2.size(), hipMemcpyHostToDevice));

    const size_t psum_bytes = (size_t)S * max_blocks_per_stream * N * sizeof(float);
    const size_t tail_bytes = (size_t)S * h->K * N * sizeof(float);
    RT_CREATE_HIP(hipMalloc(&h->d_psum, std::max<size_t>(psum_bytes, 4)));
    RT_CREATE_HIP(hipMalloc(&h->d_tail[0], tail_bytes));
    RT_CREATE_HIP(hipMalloc(&h->d_tail[1], tail_bytes));
    RT_CREATE_HIP(hipMalloc(&h->d_hot, (size_t)S * h->hot_cap * sizeof(uint2)));
    RT_CREATE_HIP(hipMalloc(&h->d_hot_count, (size_t)S * sizeof(uint32_t)));
    h->pool_cap = (int64_t)S * h->rec_cap;
    if (h->pool_cap > 0x7FFFFFFFll) h->pool_cap = 0x7FFFFFFFll;
    RT_CREATE_HIP(hipMalloc(&h->d_records, (size_t)h->pool_cap * sizeof(rt_record)));
    RT_CREATE_HIP(hipMalloc(&h->d_rec_offset, (size_t)S * sizeof(int32_t)));
    RT_CREATE_HIP(hipMalloc(&h->d_rec_count, (size_t)S * sizeof(int32_t)));
    RT_CREATE_HIP(hipMalloc(&h->d_counters, 4 * sizeof(unsigned long long)));
    RT_CREATE_HIP(hipHostMalloc(&h->h_counters, 4 * sizeof(unsigned long long)));
    RT_CREATE_HIP(hipHostMalloc(&h->h_rec_offset, (size_t)S * sizeof(int32_t)));
    RT_CREATE_HIP(hipHostMalloc(&h->h_rec_count, (size_t)S * sizeof(int32_t)));
    if (h->timing)
        for (auto &ev : h->ev) RT_CREATE_HIP(hipEventCreate(&ev));

    RT_CREATE_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(detect_sparse),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_sparse));
    RT_CREATE_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(detect_dense),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_dense));
#undef RT_CREATE_HIP
    *out = h;
    return RT_OK;
}

int rt_reset(rt_handle *h) {
    if (!h) return RT_E_INVALID;
    h->n_seg_last = -1;
    return RT_OK;
}

int rt_process(rt_handle *h, const void *iq_dev, int64_t n_samples, int64_t stream_stride) {
    if (!h) return RT_E_INVALID;
    h->call.valid = false;
    if (!iq_dev && n_samples > 0) {
        h->err = "null IQ pointer";
        return RT_E_INVALID;
    }
    if (n_samples < 0 || n_samples > h->cfg.max_samples || stream_stride < n_samples) {
        h->err = "n_samples/stream_stride out of range for this handle";
        return RT_E_INVALID;
    }
    RT_HIP(h, hipSetDevice(h->cfg.device));
    const int T = (int)(n_samples / h->N);
    if (T == 1) {
        h->err = "exactly one segment: the reference raises IndexError (times[1])";
        return RT_E_ONE_SEGMENT;
    }
    CallCtx &c = h->call;
    c = CallCtx{};
    c.iq = iq_dev;
    c.n_samples = n_samples;
    c.stream_stride = stream_stride;
    c.n_seg = T;
    c.tail_read = h->tail_cur;
    c.n_seg_last = h->n_seg_last;
    c.mode_used = (h->cfg.mode == RT_MODE_DENSE) ? RT_MODE_DENSE : RT_MODE_SPARSE;
    h->info = rt_call_info{};
    h->info.n_seg = T;
    if (T == 0) {
        // empty spectrogram: no signals; `_spectrogram_last` becomes an empty map
        RT_HIP(h, hipMemsetAsync(h->d_counters, 0, 4 * sizeof(unsigned long long), h->stream));
        RT_HIP(h, hipMemsetAsync(h->d_rec_count, 0, (size_t)h->cfg.n_streams * sizeof(int32_t), h->stream));
        RT_HIP(h, hipMemsetAsync(h->d_rec_offset, 0, (size_t)h->cfg.n_streams * sizeof(int32_t), h->stream));
        int rc = enqueue_readback(h);
        if (rc != RT_OK) return rc;
    } else {
        int rc = enqueue_analysis(h, c.mode_used == RT_MODE_DENSE);
        if (rc != RT_OK) return rc;
    }
    c.valid = true;
    h->tail_cur = 1 - c.tail_read;
    h->n_seg_last = T;
    return RT_OK;
}

int rt_process_host(rt_handle *h, const void *iq_host, int64_t n_samples, int64_t stream_stride) {
    if (!h) return RT_E_INVALID;
    if (n_samples < 0 || stream_stride < n_samples) {
        h->err = "bad n_samples/stream_stride";
        return RT_E_INVALID;
    }
    RT_HIP(h, hipSetDevice(h->cfg.device));
    const size_t bytes = (size_t)h->cfg.n_streams * (size_t)stream_stride * sizeof(cf);
    if (bytes > h->iq_stage_bytes) {
        RT_HIP(h, hipStreamSynchronize(h->stream));
        if (h->d_iq_stage) (void)hipFree(h->d_iq_stage);
        h->d_iq_stage = nullptr;
        h->iq_stage_bytes = 0;
        hipError_t e = hipMalloc(&h->d_iq_stage, bytes);
        if (e != hipSuccess) {
            h->err = std::string("IQ staging buffer: ") + hipGetErrorString(e);
            return RT_E_NOMEM;
        }
        h->iq_stage_bytes = bytes;
    }
    if (bytes) RT_HIP(h, hipMemcpyAsync(h->d_iq_stage, iq_host, bytes, hipMemcpyHostToDevice, h->stream));
    return rt_process(h, h->d_iq_stage, n_samples, stream_stride);
}

int rt_extract(rt_handle *h, const float *spec_dev, int32_t n_seg, int32_t n_bins, const float *last_dev,
               int32_t n_seg_last) {
    if (!h) return RT_E_INVALID;
    h->call.valid = false;
    if (n_seg < 0 || n_bins < 1 || (n_seg > 0 && !spec_dev) || (last_dev && n_seg_last < 0)) {
        h->err = "bad spectrogram arguments";
        return RT_E_INVALID;
    }
    if (n_seg == 1) {
        h->err = "exactly one segment: the reference raises IndexError (times[1])";
        return RT_E_ONE_SEGMENT;
    }
    if ((int64_t)n_seg * n_bins > 0x7FFFFFFFll) {
        h->err = "spectrogram too large";
        return RT_E_UNSUPPORTED;
    }
    RT_HIP(h, hipSetDevice(h->cfg.device));
    CallCtx &c = h->call;
    c = CallCtx{};
    c.n_seg = n_seg;
    c.is_extract = true;
    c.mode_used = RT_MODE_DENSE;
    h->info = rt_call_info{};
    h->info.n_seg = n_seg;
    RT_HIP(h, hipMemsetAsync(h->d_counters, 0, 4 * sizeof(unsigned long long), h->stream));
    if (n_seg == 0) {
        RT_HIP(h, hipMemsetAsync(h->d_rec_count, 0, (size_t)h->cfg.n_streams * sizeof(int32_t), h->stream));
        RT_HIP(h, hipMemsetAsync(h->d_rec_offset, 0, (size_t)h->cfg.n_streams * sizeof(int32_t), h->stream));
    } else {
        DetectArgs a = make_detect_args(h, n_seg, n_bins, last_dev ? n_seg_last : -1);
        a.dp.tail_cols = last_dev ? n_seg_last : 0;
        a.prev = last_dev;
        a.prev_cols = last_dev ? n_seg_last : 0;
        a.spec = spec_dev;
        a.psum = nullptr;  // caller-supplied map: the kernel sums the rows itself
        if (h->timing) RT_HIP(h, hipEventRecord(h->ev[0], h->stream));
        if (h->timing) RT_HIP(h, hipEventRecord(h->ev[1], h->stream));
        hipLaunchKernelGGL(detect_dense, dim3(h->cfg.n_streams), dim3(kDetBlock), h->lds_dense, h->stream, a);
        RT_HIP(h, hipGetLastError());
        if (h->timing) RT_HIP(h, hipEventRecord(h->ev[2], h->stream));
    }
    int rc = enqueue_readback(h);
    if (rc != RT_OK) return rc;
    c.valid = true;
    return RT_OK;
}

int rt_fetch(rt_handle *h, rt_record *out, size_t cap, size_t *n_out) {
    if (!h || !n_out) return RT_E_INVALID;
    *n_out = 0;
    if (!h->call.valid) {
        h->err = "rt_fetch without a preceding successful rt_process/rt_extract";
        return RT_E_INVALID;
    }
    RT_HIP(h, hipSetDevice(h->cfg.device));
    RT_HIP(h, hipStreamSynchronize(h->stream));
    CallCtx &c = h->call;
    unsigned long long flags = h->h_counters[2];
    h->info.n_hot = (int64_t)h->h_counters[1];
    if ((flags & kFlagHotOverflow) && !c.is_extract) {
        if (h->cfg.mode == RT_MODE_SPARSE) {
            h->err = "candidate-cell capacity exceeded (hot_capacity) in sparse mode";
            return RT_E_CAPACITY;
        }
        // dense re-run of the same buffer with the same look-back state
        c.fell_back = true;
        c.mode_used = RT_MODE_DENSE;
        int rc = enqueue_analysis(h, true);
        if (rc != RT_OK) return rc;
        RT_HIP(h, hipStreamSynchronize(h->stream));
        flags = h->h_counters[2];
    }
    if (flags & kFlagInconsistent) {
        h->err = "internal: candidate list lacks the cell preceding a run";
        return RT_E_HIP;
    }
    if (h->timing && c.n_seg > 0) {
        (void)hipEventElapsedTime(&h->info.ms_stft, h->ev[0], h->ev[1]);
        (void)hipEventElapsedTime(&h->info.ms_detect, h->ev[1], h->ev[2]);
        (void)hipEventElapsedTime(&h->info.ms_total, h->ev[0], h->ev[2]);
    }
    h->info.mode_used = c.mode_used;
    h->info.fell_back = c.fell_back ? 1 : 0;

    const int S = h->cfg.n_streams;
    size_t total = 0;
    for (int s = 0; s < S; ++s) total += (size_t)h->h_rec_count[s];
    h->info.n_records = (int64_t)total;
    *n_out = total;
    const size_t pool_used = (size_t)h->h_counters[0] <= (size_t)h->pool_cap ? (size_t)h->h_counters[0] : (size_t)h->pool_cap;
    if (total && out && cap) {
        if (pool_used > h->h_records_cap) {
            if (h->h_records) (void)hipHostFree(h->h_records);
            h->h_records = nullptr;
            h->h_records_cap = 0;
            const size_t want = std::max<size_t>(pool_used * 2, 4096);
            RT_HIP(h, hipHostMalloc(&h->h_records, want * sizeof(rt_record)));
            h->h_records_cap = want;
        }
        RT_HIP(h, hipMemcpyAsync(h->h_records, h->d_records, pool_used * sizeof(rt_record), hipMemcpyDeviceToHost,
                                 h->stream));
        RT_HIP(h, hipStreamSynchronize(h->stream));
        size_t w = 0;
        for (int s = 0; s < S && w < cap; ++s) {
            const int n = h->h_rec_count[s];
            const int off = h->h_rec_offset[s];
            for (int i = 0; i < n && w < cap; ++i) out[w++] = h->h_records[(size_t)off + i];
        }
    }
    if (flags & kFlagRecOverflow) {
        h->err = "record capacity exceeded (record_capacity); results truncated";
        return RT_E_CAPACITY;
    }
    return RT_OK;
}

int rt_spectrogram(rt_handle *h, const void *iq_dev, int64_t n_samples, int64_t stream_stride, float *spec_dev) {
    if (!h || !iq_dev || !spec_dev) return RT_E_INVALID;
    if (n_samples < 0 || n_samples > h->cfg.max_samples || stream_stride < n_samples) {
        h->err = "n_samples/stream_stride out of range for this handle";
        return RT_E_INVALID;
    }
    RT_HIP(h, hipSetDevice(h->cfg.device));
    const int T = (int)(n_samples / h->N);
    if (T == 0) return RT_OK;
    StftParams sp = make_stft_params(h, iq_dev, stream_stride, T, 0);
    sp.spec = spec_dev;
    launch_stft<2>(h, sp, h->cfg.n_streams * sp.blocks_per_stream);
    RT_HIP(h, hipGetLastError());
    RT_HIP(h, hipStreamSynchronize(h->stream));
    return RT_OK;
}

int rt_calibrate_read(rt_handle *h, const void *iq_dev, int64_t n_samples, int64_t stream_stride) {
    if (!h || !iq_dev) return RT_E_INVALID;
    if (n_samples < 0 || n_samples > h->cfg.max_samples || stream_stride < n_samples) {
        h->err = "n_samples/stream_stride out of range for this handle";
        return RT_E_INVALID;
    }
    RT_HIP(h, hipSetDevice(h->cfg.device));
    const int T = (int)(n_samples / h->N);
    if (T < 2) return RT_OK;
    StftParams sp = make_stft_params(h, iq_dev, stream_stride, T, 0);
    launch_stft<3>(h, sp, h->cfg.n_streams * sp.blocks_per_stream);
    RT_HIP(h, hipGetLastError());
    RT_HIP(h, hipStreamSynchronize(h->stream));
    return RT_OK;
}

int rt_get_call_info(rt_handle *h, rt_call_info *info) {
    if (!h || !info) return RT_E_INVALID;
    *info = h->info;
    return RT_OK;
}

int rt_dev_alloc(int32_t device, size_t bytes, void **out) {
    if (!out) return RT_E_INVALID;
    *out = nullptr;
    if (hipSetDevice(device) != hipSuccess) return fail_create(RT_E_NO_DEVICE, "hipSetDevice failed");
    hipError_t e = hipMalloc(out, bytes ? bytes : 4);
    if (e != hipSuccess) return fail_create(RT_E_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    return RT_OK;
}

int rt_dev_free(int32_t device, void *ptr) {
    if (hipSetDevice(device) != hipSuccess) return fail_create(RT_E_NO_DEVICE, "hipSetDevice failed");
    hipError_t e = hipFree(ptr);
    return e == hipSuccess ? RT_OK : fail_create(RT_E_HIP, std::string("hipFree: ") + hipGetErrorString(e));
}

int rt_dev_upload(int32_t device, void *dst_dev, const void *src_host, size_t bytes) {
    if (hipSetDevice(device) != hipSuccess) return fail_create(RT_E_NO_DEVICE, "hipSetDevice failed");
    hipError_t e = hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice);
    return e == hipSuccess ? RT_OK : fail_create(RT_E_HIP, std::string("hipMemcpy H2D: ") + hipGetErrorString(e));
}

int rt_dev_download(int32_t device, void *dst_host, const void *src_dev, size_t bytes) {
    if (hipSetDevice(device) != hipSuccess) return fail_create(RT_E_NO_DEVICE, "hipSetDevice failed");
    hipError_t e = hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost);
    return e == hipSuccess ? RT_OK : fail_create(RT_E_HIP, std::string("hipMemcpy D2H: ") + hipGetErrorString(e));
}

}  // extern "C"
