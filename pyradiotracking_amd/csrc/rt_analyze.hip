// rt_analyze.hip -- host side of the C-ABI declared in include/rt_analyze.h:
// handle, device scratch, kernel launches.  gfx950 only; fails loudly when no
// GPU is usable (there is no CPU fallback in the product path).
//
// Execution model of one handle
//   * two "slots" of per-call scratch (candidate lists, partial row sums,
//     record pool, counters, pinned mirrors), used alternately: the scan of
//     call k+1 may run while call k is still being detected / fetched;
//   * scan, detect kernels and readback of a call run in order on one stream
//     (the caller's stream if one was given).  Overlapping detect k with scan
//     k+1 on a second stream was measured and rejected: the scan's 3 waves/SIMD
//     x 151 VGPRs leave no register space for co-resident detect waves, so they
//     starve until the scan drains (profiles/r01_d_*).  What the two slots buy
//     is host-side pipelining: call k+1 is enqueued before call k is fetched,
//     so the GPU never waits for the host;
//   * three look-back tail buffers in rotation: call k reads (k-1)%3 and writes
//     k%3, so the scan of call k+1 never overwrites what detect k still reads;
//   * rt_fetch returns calls in FIFO order (at most two are in flight).
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
#include "rt_general.h"

using namespace rt;

namespace {

thread_local std::string g_create_error;
// (diagnostic builds: RT_TRACE_FETCH=1 prints every decision of fetch_one's re-run loop)
#define RT_TRACE_FETCH(h, sl, what)                                                                                                          \
    do {                                                                                                                                     \
        if (RT_DIAG_ENV("RT_TRACE_FETCH"))                                                                                                   \
            std::fprintf(stderr, "fetch seq %llu mode %d: %s | flags %llx records %llu wanted/stream %llu rec_cap %d (ran with %d) pool %lld grown %d/%d\n", \
                         (unsigned long long)(sl).call.seq, (sl).call.mode_used, what, (unsigned long long)(sl).h_counters[2],                \
                         (unsigned long long)(sl).h_counters[0], (unsigned long long)(sl).h_counters[4], (h)->rec_cap, (sl).call.rec_cap_used, \
                         (long long)(sl).pool_cap, (sl).call.cap_grown, (int)(sl).call.pool_grown);                                          \
    } while (0)
constexpr uint32_t kFlagLaneOfLargeBatch = 0x40000000u;  // rt_config.flags of a lane's own handle (internal): the whole batch has >= 1 024 streams
thread_local bool g_creating_lane = false;  // rt_create of a laned handle is creating one of its lanes

constexpr int kSlots = 2;
constexpr int kTails = 3;
constexpr int64_t kInitialPoolRecords = 4 << 20;  // pinned record pool per slot at rt_create unless rt_config.record_pool says otherwise
                                                  // (4 Mi records = 160 MiB); a call that needs more grows it (grow_pool), up to what
                                                  // n_streams * record_capacity can ever deliver
constexpr int kMaxPartial = 32;  // AUTO: up to this many overflowing streams of a batch are re-run dense on their own

struct CallCtx {
    bool pending = false;   // enqueued, not fetched yet
    uint64_t seq = 0;       // call number (0 = slot never used)
    const void *iq = nullptr;
    int64_t n_samples = 0;
    int64_t stream_stride = 0;
    int n_seg = 0;
    int tail_read = 0, tail_write = 0;
    int n_seg_last = -1;    // columns of the previous buffer (-1: none)
    int mode_used = 0;
    bool fell_back = false;
    bool is_extract = false;
    bool u8 = false;        // IQ is interleaved uint8 (RTL-SDR wire format)
    bool no_last = false;   // the slot's h_no_last flags apply (some stream was reset, rt_reset_stream)
    int n_dense_streams = 0;  // streams of this call that were re-run dense on their own (AUTO, partial fall-back)
    int pool_grown = 0;       // times the record pool was enlarged for this call and the call analysed again (fetch_one)
    int cap_grown = 0;        // times the handle's per-stream record capacity was enlarged for this call (fetch_one: grow_record_capacity)
    int rec_cap_used = 0;     // the per-stream record capacity its kernels ran with (a call in flight while ANOTHER call grew the capacity was still truncated at the old one)
    bool thr_rerun = false;   // analysed again on RT_MODE_RUNFILTER with thresholds from its own row means (once per call)
    bool abs_counted = false; // a MODE 4 / 6 scan of this call left the slot's h_abs_hot
    bool level_settled = false;  // AUTO's level bookkeeping for this call is done (fetch_one passes over a call twice: size query / peek, then delivery)
    bool ran_lin = false;     // its scans detrended by linearity (fetch_one: analysed again if the guard marks a stream)
    uint64_t sub_epoch = 0;   // rt_handle::sub_epoch when its kernels were enqueued
    // handle state before this call (restored when the call is rolled back: a later lane failed to enqueue)
    int prev_tail_cur = 0, prev_n_seg_last = -1, prev_dense_sticky = 0, prev_minsum_slot = -1;
};

struct Slot {
    uint2 *d_hot = nullptr;
    uint32_t *d_hot_count = nullptr;
    uint32_t *d_hot_seen = nullptr;
    int32_t *d_items = nullptr;       // [S][max_blocks][GPW] chunks of pass B's workgroups, then [S] their number per stream (with d_full)
    int32_t *h_hot_total = nullptr;   // pinned, [S]
    float *d_psum = nullptr;
    uint16_t *d_full = nullptr;       // [S][max_chunks][LG] chunk bits of the run-length pre-filter, then [S][L][LG] bits of chunk 0 by segment (null: not available)
    uint16_t *d_cell_hot = nullptr, *d_cell_need = nullptr;  // [S][max_seg][LG] threshold bits of every cell / the cells to emit (RT_MODE_RUNFILTER)
    uint32_t *d_abs_hot = nullptr;    // [S] cells at or above the absolute threshold per stream (StftParams::abs_hot; zero between calls)
    uint32_t *h_abs_hot = nullptr;    // pinned: their maximum over the streams, of this slot's latest MODE 4 / 6 scan
    float *d_thr_bin = nullptr, *d_thr_nat = nullptr;  // [S][N] per-bin thresholds of the exact pre-filter for this slot's call, lane order / bin order
                                      // (make_bin_thresholds; per slot: the check of call k reads them beside the scan of call k + 1)
    int min_items = 0;                // rows per stream of the minima d_chunk_min holds (= the work items per stream of the scan that wrote them; 0: none)
    uint64_t min_seq = 0;             // ... and the number of the call they are of
    uint32_t *d_chunk_min = nullptr;  // [S][items][N] float bits: per work item and bin the smallest complete-chunk sum of this slot's latest call (StftParams::chunk_min)
    int32_t *d_seg_list = nullptr;    // [S][max_seg] segments holding such cells, then [S] their number per stream and [1] the batch's total
    int32_t *h_seg_total = nullptr;   // pinned: that total, copied behind plan_runs
    rt_record *d_raw = nullptr;
    int32_t *d_raw_count = nullptr;
    unsigned long long *d_counters = nullptr;  // 4 words (atomics: device memory)
    // Results are written by the finalize/detect kernels straight into pinned, device-visible
    // host memory (no copy kernels); only the counter words are copied (32 bytes).
    unsigned long long *h_counters = nullptr;
    int32_t *h_rec_offset = nullptr, *h_rec_count = nullptr;  // [S]
    rt_record *h_records = nullptr;                            // [pool_cap]
    int64_t pool_cap = 0;                                      // records the slot's pool holds (grows on demand, grow_pool)
    int32_t *h_no_last = nullptr;                              // pinned, [S]: streams without a previous buffer in this call
    int32_t *h_overflow = nullptr;                             // pinned, [S]: set by detect_bucket for a stream whose candidate lists overflowed
    int32_t *h_incons = nullptr;                               // pinned, [S]: ... and for one in which a run lacked its preceding cell
    int32_t *h_dc_flag = nullptr;                              // pinned, [S]: set by a LIN scan for a stream whose constant offset is too large for that form (StftParams::dc_flag)
    int32_t *h_list = nullptr;                                 // pinned, [kMaxPartial]: the streams of a partial dense re-run (read by the kernels)
    unsigned long long *h_total = nullptr;                     // pinned: records allocated so far, uploaded before a partial re-run
    hipEvent_t ev_begin = nullptr, ev_first = nullptr, ev_scan = nullptr, ev_done = nullptr;  // first launch; end of the first scan; end of the scans; end of the call
    CallCtx call;
};

}  // namespace

struct rt_handle {
    rt_config cfg{};
    int R3 = 1, N = 256, LG = 16, GPW = 16;
    bool wcos = false;     // ... its window is a cosine sum of order <= 1: computed in the kernel from lin_c[0] + lin_c[1] cos(2 pi n / N) (stft_wg: WCOS)
    int big = 0;           // nperseg 8192 / 16 384: threads of the one-workgroup-per-segment scan (256 / 512; rt_scan_wg.h: stft_wg), else 0
    int QS = 0;            // nperseg 128 / 64 / 32: lanes of a lane group (8 / 4 / 2; R3 = 1 there), else 0 (rt_kernels.h: stft_scan<.., QS>)
    int K = 1;             // tail columns
    int stride = 1;        // probe stride
    int L = 32;            // segments per chunk
    int max_seg = 0;       // T for max_samples
    std::vector<float> h_thr_s;  // host copy of the per-stream thresholds (empty: rt_config's for every stream)
    int max_chunks = 0;
    bool two_level = false;  // stft_scan64 without the chunk-bit pre-filter: a stream's earliest chunks are half as long (rt_kernels.h: chunk_geometry)
    int max_blocks = 0;  // workgroups per stream at max_chunks
    hipStream_t s_scan = nullptr;
    bool own_scan_stream = false;
    hipStream_t s_detect = nullptr;  // the sparse detection's own (lower-priority) stream where the handle owns its streams, else = s_scan
    std::string err;

    bool general = false;      // nperseg is not one the fused scans cover: stft_general / stft_bluestein + detect_dense (rt_general.h), dense path only
    int log2n = 8;             // log2 of the LDS transform's length: nperseg (a power of two), or Bluestein's M
    cf *d_twg = nullptr;       // ... its twiddles W_M^j, j < M / 2
    bool bluestein = false;    // nperseg is not a power of two: Bluestein's algorithm with transforms of length gen_m >= 2 nperseg - 1
    int gen_m = 0;
    cf *d_cwin = nullptr;      // [nperseg] window * sqrt(scale) * exp(-i pi n^2 / nperseg)
    cf *d_bfilt = nullptr;     // [gen_m] transform of the chirp filter, divided by gen_m
    int n_cu = 256;            // compute units of the device (the scan's grid: launch_stft_lin)
    uint32_t *d_work = nullptr;  // the scan kernels' item counters (StftParams::work), zero between launches
    float *d_window = nullptr;
    float *d_window_t = nullptr;  // nperseg 4096: the window in lane order (StftParams::window_t)
    cf *d_tw1 = nullptr, *d_tw2 = nullptr;
    float *d_tail[kTails] = {nullptr, nullptr, nullptr};
    float *d_spec = nullptr;                   // lazily allocated dense spectrogram (shared)
    float *d_spec_part = nullptr;              // ... and one for min(S, kMaxPartial) streams (partial dense re-run)
    void *d_iq_stage[kSlots] = {nullptr, nullptr};  // for rt_process_host: one per call slot (a call's IQ must stay
    size_t iq_stage_bytes[kSlots] = {0, 0};         // in place until it is fetched -- AUTO mode may re-run it dense)
    int64_t pool_want = 0;  // records every slot's pool should hold: the largest size any call has needed so far
    int64_t pool_max = 0;   // ... and the most any call can deliver: (n_streams + partial re-runs) * record_capacity
    Slot slot[kSlots];

    int hot_cap = 8192, rec_cap = 1024, cand_cap = 32;
    bool group_detect = false;  // sparse detection of a whole stream by one wave (detect_group) is possible and wanted: many streams per launch ...
    bool group_light = true;    // ... and the streams of the call fetched last held few candidate cells on average (fetch_one); a call's launch looks at this
    bool group_forced = false;  // RT_FLAG_GROUP_DETECT: whatever the streams hold
    size_t lds_large = 0, lds_small = 0, lds_dense = 0;

    bool lin = false;      // constant detrend by linearity (cosine-sum window of order <= 1; rt_kernels.h: LIN)
    // ... except for the streams its guard has marked (StftParams::dc_flag / sub_first): per stream, for good (an SDR's offset is a
    // property of its hardware), so that a stream's results never depend on which other streams share its batch
    int32_t *d_sub_first = nullptr;        // [S] non-zero = marked
    std::vector<int32_t> h_sub_first;      // host copy
    int32_t *h_sub_list = nullptr;         // pinned, device-visible: the marked streams, ascending (n_sub of them)
    int n_sub = 0;
    uint64_t sub_epoch = 0;                // changes of the set so far (a call analysed under an older one is analysed again)
    float lin_c[3] = {0.f, 0.f, 0.f};

    // AUTO mode: three ways to analyse a buffer, cheapest first -- RT_MODE_SPARSE (candidates emitted by the scan itself),
    // RT_MODE_PREFILTER (two scan passes, rt_kernels.h; only where prefilter_ok), RT_MODE_DENSE.  A call whose candidate
    // lists overflow is re-run one level up when it is fetched; the handle then stays on that level for `dense_sticky`
    // calls before it probes the level below again (a failed probe costs a wasted scan: the interval doubles, 16 .. 1024).
    bool prefilter_ok = false;
    bool runfilter_ok = false;  // RT_MODE_RUNFILTER is possible and its scratch is allocated
    int minsum_slot = -1;       // the slot whose d_chunk_min holds the latest call's chunk minima (-1: none yet)
    int run_cells = 1;          // r: cells a plateau needs unless it runs through t = 0 (plan_runs)
    int auto_level = RT_MODE_SPARSE;
    int dense_sticky = 0;  // calls left on auto_level before the next probe
    // the most cells at or above the absolute threshold any stream had in the last call a pre-filter level analysed: more
    // than the sparse lists hold (kBuckets x hot_capacity per stream) means a probe of the sparse level cannot succeed
    uint32_t abs_hot_seen = 0;
    bool abs_hot_valid = false;
    int sticky_len = 16;
    uint64_t n_calls = 0;  // calls enqueued so far
    uint64_t test_enqueues = 0;  // laned handle: rt_process calls seen (fault injection, read once at rt_create: RT_TEST_FAIL_LANE=<lane>:<n>)
    int test_fail_lane = -1, test_fail_nth = 0;
    int tail_mode = 0;     // 0: a call's planning kernels and second scan in order on s_scan; 1: on s_detect with the detection (enqueue_analysis)
    int tail_cur = 0;      // tail buffer holding the most recent buffer's columns
    int n_seg_last = -1;
    std::vector<uint8_t> reset_pending;  // [S] streams whose look-back is dropped at the next rt_process
    bool any_reset_pending = false;
    float *d_thr_s = nullptr, *d_cal_s = nullptr;  // [S] per-stream thresholds / calibration (rt_set_stream_params)

    rt_call_info info{};
    bool timing = false;

    // cfg.lanes > 1: this handle only owns `kids` (one complete handle per stream group, each with its own
    // HIP stream) and forwards every call to them; kid k analyses streams [kid_base[k], kid_base[k + 1])
    std::vector<rt_handle *> kids;
    std::vector<int> kid_base;
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

#ifdef RT_DIAG
// The diagnostic library reads a handful of environment switches (rt_diag.h).  A committed script that still sets one that no
// longer exists would A/B two identical configurations without a word (advisor, round 5): every RT_EXP_* / RT_TEST_* variable
// this build does not know is named on stderr, once per process.
extern "C" char **environ;
void warn_unknown_switches() {
    static bool done = false;
    if (done) return;
    done = true;
    static const char *const known[] = {"RT_EXP_ONE_LEVEL", "RT_EXP_TAIL", "RT_EXP_STREAMS", "RT_EXP_TAIL_PRIO", "RT_TEST_FAIL_LANE", "RT_STAMPS_DUMP"};
    for (char **e = environ; e && *e; ++e) {
        if (std::strncmp(*e, "RT_EXP_", 7) != 0 && std::strncmp(*e, "RT_TEST_", 8) != 0) continue;
        const char *eq = std::strchr(*e, '=');
        const std::string name(*e, eq ? (size_t)(eq - *e) : std::strlen(*e));
        bool ok = false;
        for (const char *k : known) ok = ok || name == k;
        if (!ok) std::fprintf(stderr, "librt_analyze_diag: environment variable %s is set but this build reads no such switch (known: RT_EXP_ONE_LEVEL, RT_EXP_TAIL, RT_EXP_STREAMS, RT_EXP_TAIL_PRIO, RT_TEST_FAIL_LANE, RT_STAMPS_DUMP)\n", name.c_str());
    }
}
#endif

int fail_create(int code, const std::string &msg) {
    g_create_error = msg;
    return code;
}

size_t rec_lds_bytes(int rec_cap) { return (size_t)std::min(rec_cap, kDenseLdsRecords) * (8 + 8 + sizeof(rt_record)) + 16; }

// The dense extractor over `grid` streams (all of them, or the list in a.stream_list).  While the handle's record capacity fits
// the kernel's LDS staging it orders, filters and publishes its records itself; beyond that (a capacity that has grown: the
// reference appends without limit, analyze.py:449-450) the list is staged in the streams' raw-record areas and finalize_records,
// the sparse path's last kernel, finishes the call.
void launch_detect_dense(rt_handle *h, int grid, hipStream_t st, const DetectArgs &a);

int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

// `items` work items (one per stream and group of GPW chunks).  Every mode but the selective pass runs them on a grid
// that just fills the chip: CUs x the workgroups of this instantiation a CU holds (its __launch_bounds__), each
// workgroup drawing further items from p.work (rt_kernels.h: "Work items") -- more workgroups than that would only
// queue in the dispatcher and find the counter exhausted.
template <int MODE, bool U8, bool LIN>
void launch_stft_lin(rt_handle *h, const StftParams &p, int items, hipStream_t st) {
    if (h->big) {
        // nperseg 8192 / 16 384: one workgroup per item (a chunk of one stream), a segment per step (rt_scan_wg.h)
        if constexpr (MODE <= 2) {
            if (h->big == 256) {
                if (h->wcos) hipLaunchKernelGGL((stft_wg<256, MODE, U8, true>), dim3(items), dim3(256), wg_lds_bytes(256), st, p);
                else hipLaunchKernelGGL((stft_wg<256, MODE, U8, false>), dim3(items), dim3(256), wg_lds_bytes(256), st, p);
            } else {
                if (h->wcos) hipLaunchKernelGGL((stft_wg<512, MODE, U8, true>), dim3(items), dim3(512), wg_lds_bytes(512), st, p);
                else hipLaunchKernelGGL((stft_wg<512, MODE, U8, false>), dim3(items), dim3(512), wg_lds_bytes(512), st, p);
            }
        }
        return;  // (the other modes do not exist at these sizes: rt_create refuses them)
    }
    if (scan_wave64(h->R3)) {
        // nperseg 4096: one wave per segment, items drawn per wave; one 8-wave workgroup (all of a CU's LDS) per CU
        const int wgs = std::min((items + kW64Waves - 1) / kW64Waves, h->n_cu);
        hipLaunchKernelGGL((stft_scan64<MODE, U8, LIN>), dim3(wgs), dim3(kW64Block), 0, st, p);
        return;
    }
    const int blk = scan_block(h->R3);
    // workgroups a CU holds: three waves per SIMD by registers (four for the leaner uint8 / RT_WG4 instantiations), and at
    // nperseg 256 twelve one-wave workgroups by LDS whatever the registers allow
    const int per_cu = scan_dma(h->R3, U8, h->QS) ? 2 : std::min(((h->R3 <= RT_WG4_MAX_R3 || (U8 && h->R3 == 1)) ? 4 : 3) * (kBlock / blk), blk == 64 ? 12 : 4);
    // (A/B on one box, whole path, profiles/r03_h_persistent_ab.txt: config 3 one lane 662 k -> 684 k MS/s, config 5 share +1 %)
    const bool persist = scan_persistent(h->R3, MODE);
    const int blocks = persist ? std::min(items, h->n_cu * per_cu) : items;
    switch (h->QS ? -h->QS : h->R3) {
        case -8: hipLaunchKernelGGL((stft_scan<1, MODE, U8, LIN, 8>), dim3(blocks), dim3(blk), 0, st, p); break;  // nperseg 128
        case -4: hipLaunchKernelGGL((stft_scan<1, MODE, U8, LIN, 4>), dim3(blocks), dim3(blk), 0, st, p); break;  // 64
        case -2: hipLaunchKernelGGL((stft_scan<1, MODE, U8, LIN, 2>), dim3(blocks), dim3(blk), 0, st, p); break;  // 32
        case 1: hipLaunchKernelGGL((stft_scan<1, MODE, U8, LIN>), dim3(blocks), dim3(blk), 0, st, p); break;
        case 2: hipLaunchKernelGGL((stft_scan<2, MODE, U8, LIN>), dim3(blocks), dim3(blk), 0, st, p); break;
        case 4: hipLaunchKernelGGL((stft_scan<4, MODE, U8, LIN>), dim3(blocks), dim3(blk), 0, st, p); break;
        case 8: hipLaunchKernelGGL((stft_scan<8, MODE, U8, LIN>), dim3(blocks), dim3(blk), 0, st, p); break;
#if !RT_WAVE64_4096
        default: hipLaunchKernelGGL((stft_scan<16, MODE, U8, LIN>), dim3(blocks), dim3(blk), 0, st, p); break;
#else
        default: break;  // (nperseg 4096 is stft_scan64's, above)
#endif
    }
}

// MODE 3 (load stream only) has no detrend: one instantiation
// uint8 input keeps the subtract-first form: quantised samples make exact cancellations real (a saturated segment is
// constant, x - mean is exactly zero and so is every cell of it in the reference -> std = NaN over a plateau that holds
// one; the linearity form leaves a residue 140 dB under the offset there -- found by the round-2 soak)
template <int MODE, bool U8 = false>
void launch_stft(rt_handle *h, const StftParams &p, int blocks, hipStream_t st) {
    if (h->lin && MODE != 3 && !U8) {
        if (p.stream_list || h->n_sub == 0 || !p.sub_first) {
            // (a launch over a list of its own -- AUTO's dense re-run of a few streams -- keeps the linearity form for all of them: its
            // dense spectrogram is indexed by position in that list, which a second launch over a sub-list cannot address)
            StftParams q = p;
            if (p.stream_list) q.sub_first = nullptr;
            launch_stft_lin<MODE, U8, (MODE != 3 && !U8)>(h, q, blocks, st);
        } else {
            // the streams the guard of that form has marked (StftParams::dc_flag): the first launch leaves them alone, the
            // subtract-first instantiation takes them, by list
            launch_stft_lin<MODE, U8, (MODE != 3 && !U8)>(h, p, blocks, st);
            StftParams q = p;
            q.sub_first = nullptr;
            q.dc_flag = nullptr;
            q.spec_by_stream = 1;
            q.stream_list = h->h_sub_list;
            q.n_streams = h->n_sub;
            launch_stft_lin<MODE, U8, false>(h, q, h->n_sub * p.blocks_per_stream, st);
        }
    } else {
        launch_stft_lin<MODE, U8, false>(h, p, blocks, st);
    }
}

// the general transform (rt_general.h): the dense spectrogram of an nperseg the fused scans do not cover (the caller runs
// row_sums_dense over the map)
void launch_general(rt_handle *h, const void *iq, int64_t stream_stride, int n_seg, float *spec, float *tail, bool u8) {
    if (h->bluestein) {
        BluesteinParams b{};
        b.iq = iq;
        b.stream_stride = stream_stride;
        b.n_streams = h->cfg.n_streams;
        b.n_seg = n_seg;
        b.nperseg = h->N;
        b.m = h->gen_m;
        b.log2m = h->log2n;
        b.tail_cols = h->K;
        b.cwin = h->d_cwin;
        b.bfilt = h->d_bfilt;
        b.tw = h->d_twg;
        b.spec = spec;
        b.tail = tail;
        const int blocks = h->cfg.n_streams * n_seg;
        const size_t lds = (size_t)h->gen_m * sizeof(cf);
        // (groups of a thread per trip of a double stage: rt_general.h, lds_fft_stages -- M / 4 a multiple of 256 U)
        // threads per workgroup and groups of a thread per trip of a double stage (rt_general.h, lds_fft_stages: M / 4 a multiple of threads x U).
        // From M = 4096 on 512 threads, at 16 384 (one workgroup per CU) 1 024: more waves to cover the LDS latency of the stages
        // where LDS leaves room for one or two workgroups per CU (16 384: 512 threads 27.4 k MS/s at nperseg 6000, 1 024 34 k).
#define RT_BLU(U_, B_)                                                                                                            \
    do {                                                                                                                          \
        if (u8) hipLaunchKernelGGL((stft_bluestein<true, U_, B_>), dim3(blocks), dim3(B_), lds, h->s_scan, b);                     \
        else hipLaunchKernelGGL((stft_bluestein<false, U_, B_>), dim3(blocks), dim3(B_), lds, h->s_scan, b);                       \
    } while (0)
        if (h->gen_m >= 16384) RT_BLU(4, 1024);
        else if (h->gen_m >= 8192) RT_BLU(4, 512);  // (8 groups x 256 threads: 31.8 k MS/s at nperseg 3000, this 41.6 k)
        else if (h->gen_m >= 4096) RT_BLU(2, 512);  // (4 x 256: 47.9 k at nperseg 1500, this 51.2 k)
        else if (h->gen_m >= 2048) RT_BLU(2, 256);  // (1 x 512: 56.7 k at nperseg 1000, this 72 k -- enough workgroups per CU as it is)
        else RT_BLU(1, 256);
#undef RT_BLU
        return;
    }
    GeneralParams g{};
    g.iq = iq;
    g.stream_stride = stream_stride;
    g.n_streams = h->cfg.n_streams;
    g.n_seg = n_seg;
    g.nperseg = h->N;
    g.log2n = h->log2n;
    g.segs_per_block = std::min(64, std::max(1, 1024 / h->N));  // (N / 4 four-element groups per segment and double stage: every thread has one at N <= 1024)
    g.tail_cols = h->K;
    g.window = h->d_window;
    g.tw = h->d_twg;
    g.spec = spec;
    g.tail = tail;
    const int blocks = h->cfg.n_streams * ((n_seg + g.segs_per_block - 1) / g.segs_per_block);
    const size_t lds = (size_t)g.segs_per_block * h->N * sizeof(cf);
    if (u8) hipLaunchKernelGGL((stft_general<true>), dim3(blocks), dim3(kGeneralBlock), lds, h->s_scan, g);
    else hipLaunchKernelGGL((stft_general<false>), dim3(blocks), dim3(kGeneralBlock), lds, h->s_scan, g);
}

// cells a run must have to pass the duration gate unless it runs through t = 0 (see rt_create)
long long min_run_cells(const rt_config &cfg, int N) {
    const double hop = seg_time(1, N, cfg.sample_rate) - seg_time(0, N, cfg.sample_rate);
    const double cells = cfg.min_duration_s * (1.0 - 1e-9) / hop;
    return (long long)std::ceil(std::min(cells, 1.0e9)) - 1;
}
long long min_run_cells(const rt_handle *h) { return min_run_cells(h->cfg, h->N); }

// segments per chunk for a handle of `n_streams` streams (for a laned handle: of all lanes together -- the lanes take the
// parent's choice, so that a stream's row sums are added in the same order however the batch is split into lanes)
int choose_chunk(const rt_config &cfg, int R3, int QS, int n_streams, int n_seg) {
    if (cfg.segs_per_chunk > 0) return cfg.segs_per_chunk;
    if (cfg.nperseg >= 8192) {
        // stft_wg: one chunk per workgroup, 512 workgroup slots on the chip (two per CU at 8192; 256 at 16 384).  A workgroup pays
        // ~2 steps on top of its L (tables, the first segment's round trip with nothing to overlap it, the row sums' stores): chunks of
        // about 40 segments where the batch fills the chip eight times over, shorter ones -- down to 8 -- for small batches; then the
        // length that leaves no short last chunk.
        int L = 40;
        while (L > 8 && (int64_t)n_streams * ((n_seg + L - 1) / L) < 8 * 512) L -= 8;
        const int chunks = std::max(1, (n_seg + L - 1) / L);
        return std::max(1, (n_seg + chunks - 1) / chunks);
    }
    const int N = QS ? 16 * QS : 256 * R3, GPW = scan_block(R3) / (QS ? QS : 16 * R3);
    // enough workgroups to fill 256 CUs several times over, halo overhead <= 1/L
    int L = 32;
    // ... but where the run-length pre-filter is possible with chunks of 32 (minimum duration >= 64 hops) the chunks
    // stay that long for small batches too: its selectivity is p^L (a small batch is launch-bound anyway)
    // (whatever the mode: the chunk length sets the order of the row sums' partial sums, and the modes return the same bits)
    const bool keep_long = 2ll * L - 1 <= min_run_cells(cfg, N) && n_seg >= 2 * L;
    while (L > 4 && !keep_long) {
        const int64_t chunks = (n_seg + L - 1) / L;
        const int64_t blocks = (int64_t)n_streams * ((chunks + GPW - 1) / GPW);
        if (blocks >= 2048) break;
        L >>= 1;
    }
    if (L == 32 && R3 < 4 && n_seg > 32 && !keep_long) {
        // nperseg <= 512, a batch that fills the chip, no chunk bits to serve: a workgroup holds GPW chunks and its lane groups walk in
        // step, so a stream costs (workgroups) x L steps whatever its last workgroup holds -- 1 171 segments (the reference's default
        // geometry) are 37 chunks of 32 in three workgroups of 16, eleven lane groups idle, or 47 chunks of 25 in the same three, one
        // idle: 22 % fewer steps (551 -> 636 k MS/s).  Among the lengths 20 .. 32 the one with the fewest steps, a step of overhead per
        // workgroup, and 2 % against multiples of eight (a wave's four lane groups read four chunks L x 2 KiB apart: 32- and 64-KiB
        // strides are the slowest per step in every sweep).  (Not a function of the number of streams, like the rules below.)
        // Where the chunk bits exist (config 2 / 4 geometry) the chunks stay 32 long: at config 2 a length of 25 (20 full workgroups
        // per stream instead of 15.6) makes the scan alone 1.5 - 5.5 % faster on four boxes and the uint8 path 2 %, the whole path with
        // two lanes the same, and the chunk-bit and exact pre-filter levels 2 - 7 % slower (shorter chunks are less selective, more
        // workgroups in the second scan); config 4 (2 048 segments = four full workgroups) is fastest at 32 anyway --
        // profiles/r04_q_chunk_length_sweep_nperseg256.txt.
        double best = 0.0;
        for (int cand = 20; cand <= 32; ++cand) {
            const int64_t chunks = (n_seg + cand - 1) / cand;
            const int64_t wgs = (chunks + GPW - 1) / GPW;
            const double cost = (double)wgs * (cand + 1.0) * (cand % 8 == 0 ? 1.02 : 1.0);
            if (best == 0.0 || cost < best * (1.0 - 1e-9)) {
                best = cost;
                L = cand;
            }
        }
    }
    if (L == 32 && !keep_long && R3 >= 4) {
        // nperseg >= 1024, a batch that fills the chip: the chunk length is chosen by what a workgroup costs.  All lane
        // groups of a workgroup take L steps (a last chunk that is short leaves its group idle), and a workgroup pays
        // c0 steps on top: tables into LDS, the first segment's HBM round trip with nothing to overlap it, the halo
        // step (nperseg 4096), the row-sum epilogue.  Measured (profiles/r03_a_chunk_length_sweep.txt, one lane):
        // nperseg 4096, 781 segments: L = 32 -> 71 (11 chunks, none short) takes 5.8 - 6.7 % less time at 1 024 and at
        // 4 096 streams, L = 52 (a last chunk of one segment) 1.6 % more; that fits c0 = 3.8.  nperseg 1024, 2 343 segments,
        // four chunks to a workgroup: L = 28 / 31 (84 / 76 chunks: 21 / 19 full workgroups) take 4 % less than 32 (19
        // workgroups, three chunk slots idle), 36 and 64 more: c0 = 2.  (No term for the end of the launch: the choice
        // must not depend on the number of streams, or shards of one population would add their row sums in different orders.)
        // nperseg 2048 (no halo step: c0 = 2.8; 512 streams x 1 000 segments): L = 72 (7 workgroups per stream) 2 % less
        // than 32, but 48 / 62 / 77 take 4 - 13 % more -- with 3 584 workgroups the launch is under five rounds of the
        // chip's 768 slots and its last round counts: the search keeps at least eight rounds.
        const double c0 = R3 >= 16 ? 3.8 : R3 >= 8 ? 2.8 : 2.0;
        const int lo = 24, hi = R3 >= 8 ? 80 : 40;
        double best = 0.0;
        for (int cand = lo; cand <= hi; ++cand) {
            const int64_t chunks = (n_seg + cand - 1) / cand;
            const int64_t wgs = (chunks + GPW - 1) / GPW;
            if ((int64_t)n_streams * wgs < 8 * 768) break;  // (nothing qualifies: L stays 32)
            const double cost = (double)wgs * (cand + c0);
            if (best == 0.0 || cost < best * (1.0 - 1e-9)) {
                best = cost;
                L = cand;
            }
        }
    }
    return L;
}

int key_tbits(int n_seg) {
    int t = 1;
    while ((1ll << t) < (long long)n_seg) ++t;
    return t;
}

StftParams make_stft_params(rt_handle *h, Slot &sl, const void *iq, int64_t stream_stride, int n_seg, int tail_write) {
    StftParams p{};
    p.iq = iq;
    p.stream_stride = stream_stride;
    p.n_streams = h->cfg.n_streams;
    p.n_seg = n_seg;
    p.segs_per_chunk = h->L;
    p.chunks = (n_seg + h->L - 1) / h->L;
    if (h->two_level) {
        int n_short = 0, short_len = 0, chunks = 0;
        chunk_geometry(n_seg, h->L, true, &n_short, &short_len, &chunks);
        p.short_chunks = n_short;
        p.short_len = short_len;
        p.chunks = chunks;
    }
    p.blocks_per_stream = (p.chunks + h->GPW - 1) / h->GPW;
    p.tail_cols = h->K;
    p.work = h->d_work;
    p.window = h->d_window;
    p.window_t = h->d_window_t;
    p.tw1 = h->d_tw1;
    p.tw2 = h->d_tw2;
    p.scale = h->cfg.scale;
    p.thr = h->cfg.threshold;
    for (int i = 0; i < 3; ++i) p.lin_c[i] = h->lin_c[i];
    p.thr_s = h->d_thr_s;
    p.psum = sl.d_psum;
    p.tail = h->d_tail[tail_write];
    p.spec = h->d_spec;
    p.hot = sl.d_hot;
    p.hot_count = sl.d_hot_count;
    p.hot_cap = h->hot_cap;
    p.tbits = key_tbits(n_seg);
    p.full = sl.d_full;
    p.first = sl.d_full ? sl.d_full + (size_t)h->cfg.n_streams * h->max_chunks * h->LG : nullptr;
    p.item_chunks = sl.d_items;
    p.item_count = sl.d_items ? sl.d_items + (size_t)h->cfg.n_streams * h->max_blocks * h->GPW : nullptr;
    p.chunk_min = nullptr;  // (only the exact pre-filter's own scans keep chunk minima: enqueue_analysis)
    p.abs_hot = nullptr;
    p.thr_bin = nullptr;
    p.cell_hot = sl.d_cell_hot;
    p.cell_need = sl.d_cell_need;
    p.seg_list = sl.d_seg_list;
    p.seg_count = sl.d_seg_list ? sl.d_seg_list + (size_t)h->cfg.n_streams * h->max_seg : nullptr;
    p.dc_flag = h->lin ? sl.h_dc_flag : nullptr;
    p.sub_first = h->lin ? h->d_sub_first : nullptr;
    p.dc_limit = 1.0e6f * (float)h->N * (float)h->N * (float)h->cfg.sample_rate;  // 60 dB over the quietest bin's per-sample power
    p.dc_limit2 = 100.0f * (float)h->N * (float)h->cfg.sample_rate;               // ... and 20 dB over everything else in those segments
    return p;
}

DetectArgs make_detect_args(rt_handle *h, Slot &sl, int n_seg, int n_bins, int n_seg_last) {
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
    a.hot = sl.d_hot;
    a.hot_count = sl.d_hot_count;
    a.hot_count_rw = sl.d_hot_count;
    a.large_any = sl.d_hot_seen;  // (one word per stream of the array's S * 16; behind them detect_group's work list, then its counter)
    if (h->group_detect && (h->group_forced || h->group_light)) {
        a.work_list = reinterpret_cast<int32_t *>(sl.d_hot_seen) + h->cfg.n_streams;
        a.work_count = a.work_list + (size_t)h->cfg.n_streams * kQuarters;
    }
    a.hot_total = sl.h_hot_total;
    a.lds_cells = next_pow2(std::max(h->hot_cap, 64));
    a.cand_cap = h->cand_cap;
    a.hot_cap = h->hot_cap;
    a.tbits = key_tbits(n_seg);
    a.raw = sl.d_raw;
    a.raw_count = sl.d_raw_count;
    a.psum = sl.d_psum;
    a.records = sl.h_records;
    a.pool_cap = sl.pool_cap;
    a.rec_cap = h->rec_cap;
    a.rec_offset = sl.h_rec_offset;
    a.rec_count = sl.h_rec_count;
    a.counters = sl.d_counters;
    a.host_counters = sl.h_counters;
    a.thr_s = h->d_thr_s;
    a.cal_s = h->d_cal_s;
    a.no_last = sl.call.no_last ? sl.h_no_last : nullptr;
    a.stream_overflow = sl.h_overflow;
    a.stream_incons = sl.h_incons;
    return a;
}

void launch_detect_dense(rt_handle *h, int grid, hipStream_t st, const DetectArgs &a) {
    if (h->rec_cap <= kDenseLdsRecords) {
        hipLaunchKernelGGL(detect_dense<false>, dim3(grid), dim3(kDetBlock), h->lds_dense, st, a);
    } else {
        hipLaunchKernelGGL(detect_dense<true>, dim3(grid), dim3(kDetBlock), 0, st, a);
        hipLaunchKernelGGL(finalize_records, dim3(grid), dim3(256), 0, st, a);
    }
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

// calls without a detect kernel (empty spectrogram): counter words -> pinned host memory by a copy
int enqueue_readback(rt_handle *h, Slot &sl, hipStream_t st) {
    RT_HIP(h, hipMemcpyAsync(sl.h_counters, sl.d_counters, kCounterWords * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    return RT_OK;
}

// the level above / below `mode` in AUTO's order (rt_core.h: level_up / level_down / level_rank)
static_assert(kAutoDense == RT_MODE_DENSE && kAutoSparse == RT_MODE_SPARSE && kAutoPrefilter == RT_MODE_PREFILTER && kAutoRunfilter == RT_MODE_RUNFILTER,
              "rt_core.h mirrors rt_mode");
int level_up(const rt_handle *h, int mode) { return level_up(AutoLevels{h->prefilter_ok, h->runfilter_ok}, mode); }
int level_down(const rt_handle *h, int mode) { return level_down(AutoLevels{h->prefilter_ok, h->runfilter_ok}, mode); }

// (`st`: the handle's scan stream, or -- the selective second scans of the pre-filter levels -- the stream of what follows a call's first scan)
template <int MODE>
void launch_scan(rt_handle *h, const StftParams &sp, int blocks, bool u8, hipStream_t st = nullptr) {
    if (!st) st = h->s_scan;
    if (u8) launch_stft<MODE, true>(h, sp, blocks, st); else launch_stft<MODE>(h, sp, blocks, st);
}

// enqueue scan + detect + readback for the call described by sl.call, analysed the way `mode` says.
// `second_pass_only`: RT_MODE_PREFILTER for a call whose RT_MODE_SPARSE attempt has just overflowed -- that scan
// wrote the chunk bits, row sums and tail columns already, only the selective pass and the detection are repeated.
// `own_means`: RT_MODE_RUNFILTER for a call whose per-bin thresholds have just failed their check -- the thresholds of
// the re-run come from the row means the failed scan left (make_bin_thresholds_from_means).
int enqueue_analysis(rt_handle *h, Slot &sl, int mode, bool *launched = nullptr, bool second_pass_only = false, bool own_means = false) {
    sl.call.rec_cap_used = h->rec_cap;
    const CallCtx &c = sl.call;
    if (launched) *launched = false;
    if (h->general) {
        // any other power-of-two nperseg: the general transform into the dense map, then the dense extractor (which sums the rows itself)
        int rc = ensure_dense_spec(h);
        if (rc != RT_OK) return rc;
        RT_HIP(h, hipEventRecord(sl.ev_begin, h->s_scan));
        if (launched) *launched = true;
        launch_general(h, c.iq, c.stream_stride, c.n_seg, h->d_spec, h->d_tail[c.tail_write], c.u8);
        {
            const int64_t cells = (int64_t)h->cfg.n_streams * h->N;
            hipLaunchKernelGGL(row_sums_dense, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, h->s_scan, h->d_spec, sl.d_psum, h->cfg.n_streams, c.n_seg, h->N);
        }
        RT_HIP(h, hipGetLastError());
        RT_HIP(h, hipEventRecord(sl.ev_scan, h->s_scan));
        DetectArgs a = make_detect_args(h, sl, c.n_seg, h->N, c.n_seg_last);
        a.prev = h->d_tail[c.tail_read];
        a.prev_cols = h->K;
        a.spec = h->d_spec;
        a.psum = sl.d_psum;  // one partial row per stream (row_sums_dense)
        a.chunks = 1;
        launch_detect_dense(h, h->cfg.n_streams, h->s_scan, a);
        RT_HIP(h, hipGetLastError());
        RT_HIP(h, hipEventRecord(sl.ev_done, h->s_scan));
        return RT_OK;
    }
    StftParams sp = make_stft_params(h, sl, c.iq, c.stream_stride, c.n_seg, c.tail_write);
    if (sp.chunks > h->max_chunks) {
        h->err = "internal: chunk count exceeds scratch";
        return RT_E_INVALID;
    }
    const int blocks = h->cfg.n_streams * sp.blocks_per_stream;
    const int S = h->cfg.n_streams;
    const bool dense = (mode == RT_MODE_DENSE);
    // hot_count / raw_count and the four counter words are left zero by their last readers
    // (detect_bucket<true>, finalize_records / detect_dense: close_call)
    if (dense) {
        int rc = ensure_dense_spec(h);
        if (rc != RT_OK) return rc;
        sp.spec = h->d_spec;
    }
    if ((mode == RT_MODE_PREFILTER && !sl.d_full) || (mode == RT_MODE_RUNFILTER && !sl.d_cell_hot)) {
        h->err = "internal: pre-filter without its scratch";
        return RT_E_INVALID;
    }
    if (!second_pass_only) RT_HIP(h, hipEventRecord(sl.ev_begin, h->s_scan));
    if (launched) *launched = true;
    sl.call.ran_lin = h->lin && !c.u8;
    sl.call.sub_epoch = h->sub_epoch;
    const int slot_index = (int)(&sl - h->slot);
    if (mode == RT_MODE_RUNFILTER) {
        // per-bin thresholds from the latest chunk minima (the previous call's; on a re-run this call's own), before they are reset
        const int64_t cells = (int64_t)S * h->N;
        if (own_means) {
            hipLaunchKernelGGL(make_bin_thresholds_from_means, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, h->s_scan, sl.d_psum, sp.blocks_per_stream,
                               c.n_seg, sl.d_thr_bin, sl.d_thr_nat, S, h->R3, h->LG, h->cfg.snr_threshold);
        } else {
        // Chunk minima are kept by this level's own scans only (round 5: the sparse scans of an AUTO handle paid for them in every
        // item's epilogue -- config 3 +1.8 % per launch -- for the one call in thousands that climbs here).  None on hand -- the handle's
        // first call on this level, or the first after calls on other levels (the minima must be the previous call's): a scan of this
        // buffer provides them (its bits, taken with the absolute threshold alone, are overwritten by the scan proper below).
        // "On hand" means: of the buffer immediately before this one (or of this very buffer: a call analysed again on this level).  An
        // AUTO handle that comes back here after thousands of sparse calls would otherwise build its thresholds from the minima of an
        // arbitrarily old buffer -- too high (a stale-threshold re-analysis and a host sync) or too low (an unselective filter, which
        // pushes AUTO on to the dense path for sticky_len calls): advisor, round 5.
        sp.chunk_min = sl.d_chunk_min;
        int src = h->minsum_slot;
        const bool on_hand = src >= 0 && h->slot[src].min_items > 0 && (h->slot[src].min_seq + 1 == c.seq || h->slot[src].min_seq == c.seq);
        if (!on_hand) {
            launch_scan<6>(h, sp, blocks, c.u8);
            sl.min_items = sp.blocks_per_stream;
            sl.min_seq = sl.call.seq;
            src = slot_index;
            // (the LATEST buffer's minima set the next call's thresholds: a call analysed again from rt_fetch keeps that place for a later one)
            if (h->minsum_slot < 0 || sl.min_seq >= h->slot[h->minsum_slot].min_seq) h->minsum_slot = slot_index;
        }
        const Slot &ps = h->slot[src];
        hipLaunchKernelGGL(make_bin_thresholds, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, h->s_scan, ps.d_chunk_min, ps.min_items, sl.d_thr_bin,
                           sl.d_thr_nat, S, h->R3, h->LG, h->L * minsum_group(h->L, std::min(h->GPW, 16)), h->cfg.snr_threshold);
        }
    }
    if (mode == RT_MODE_RUNFILTER && !second_pass_only) {
        sp.chunk_min = sl.d_chunk_min;
        sl.min_items = sp.blocks_per_stream;  // (every item of the scan below writes its row of minima: nothing to reset)
        sl.min_seq = sl.call.seq;
        // the LATEST buffer's minima set the next call's thresholds: a call analysed again from rt_fetch (level-up, stale
        // thresholds, pool growth, the detrend guard) while a later one is in flight must not take that place back
        if (h->minsum_slot < 0 || sl.min_seq >= h->slot[h->minsum_slot].min_seq) h->minsum_slot = slot_index;
    }
    // `sd`: the stream of the call's detection (and, experimentally, of more of what follows its first scan).  Where the handle has
    // a second stream (rt_create: nperseg <= 512 without lanes) the first scan of the NEXT call, ready at the same moment on s_scan,
    // takes the chip while this call's detection runs beside it (config 2 one lane 0.79 -> 0.73 ms per step, profiles/r04_e_*).
    // What comes BETWEEN a call's two scans on the pre-filter levels -- planning kernels, the selective second scan -- stays in
    // order on s_scan (tail_mode 0): moved to `sd` as well (tail_mode 1, a diagnostic build's RT_EXP_TAIL=C) the second scan is
    // dispatched behind the next call's chip-filling first scan and gets no workgroup slot until that scan has handed out its last
    // workgroup -- it ends with it, 1.9 ms instead of 0.23, the detection behind it is exposed, rt_fetch returns a scan late and
    // the host enqueues late: 2.33 - 2.49 ms per step against 2.26 - 2.45 in order at the reference's default geometry, whatever
    // the stream's priority (profiles/r05_b_tail_on_second_stream.txt).  Safe by the slots: everything the tail reads or writes is
    // its call's slot's (per-bin thresholds included), the look-back tail it reads is two rotations from the one the next scan
    // writes, and a scan that reuses the slot waits for ev_done (claim_slot).  The dense path stays in order on the scan's stream:
    // its spectrogram is one per handle.
    hipStream_t sd = dense ? h->s_scan : h->s_detect;
    hipStream_t s2 = (h->tail_mode >= 1 && !dense) ? sd : h->s_scan;  // the stream of the planning kernels and the second scan
    auto behind_first_scan = [&]() -> int {
        if (s2 != h->s_scan) {
            RT_HIP(h, hipEventRecord(sl.ev_first, h->s_scan));
            RT_HIP(h, hipStreamWaitEvent(s2, sl.ev_first, 0));
        }
        return RT_OK;
    };
    if (dense) {
        launch_scan<1>(h, sp, blocks, c.u8);
    } else if (mode == RT_MODE_RUNFILTER) {
        // threshold bits of every cell (+ row sums, tail) -> cells of runs long enough -> only their segments again
        sp.thr_bin = sl.d_thr_bin;
        sp.abs_hot = sl.d_abs_hot;
        launch_scan<6>(h, sp, blocks, c.u8);
        sp.thr_bin = nullptr;
        sp.abs_hot = nullptr;
        { const int rc = behind_first_scan(); if (rc != RT_OK) return rc; }
        {
            // one launch behind the scan: the largest per-stream count of cells over the absolute threshold (AUTO's probes), the
            // check of the per-bin thresholds against this buffer's row means, the segment counters back to zero
            const int64_t cells = (int64_t)S * h->N;
            hipLaunchKernelGGL(after_bit_scan, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, s2, sl.d_abs_hot, sl.h_abs_hot, sl.d_thr_nat, sl.d_psum, S, h->N,
                               sp.blocks_per_stream, c.n_seg, h->cfg.snr_threshold, sl.h_overflow, sl.d_counters, kFlagHotOverflow | kFlagThrStale,
                               const_cast<int32_t *>(sp.seg_count));
        }
        sl.call.abs_counted = true;
        {
            // (a run needs r cells: more than the buffer holds means "only the run through t = 0", which r = n_seg + 1 says as well)
            const int r = (int)std::min<long long>(h->run_cells, (long long)c.n_seg + 1);
            const int tile = plan_tile_rows(c.n_seg, h->LG, r);
            const int tpw = 64 / (h->LG / 4);
            const int waves = (c.n_seg + tpw * tile - 1) / (tpw * tile);
            const size_t plan_lds = ((size_t)tpw * tile + 3) & ~(size_t)3;
            auto *kern = r <= 16 ? plan_runs<4> : r <= 256 ? plan_runs<8> : plan_runs<16>;
            hipLaunchKernelGGL(kern, dim3(S, waves), dim3(64), plan_lds, s2, sp.cell_hot, sl.d_cell_need,
                               sl.d_seg_list, const_cast<int32_t *>(sp.seg_count), c.n_seg, h->LG, r, tile);
        }
        launch_scan<7>(h, sp, blocks, c.u8, s2);
    } else if (mode == RT_MODE_PREFILTER) {
        if (!second_pass_only) {
            sp.abs_hot = sl.d_abs_hot;
            launch_scan<4>(h, sp, blocks, c.u8);
            sp.abs_hot = nullptr;
        }
        { const int rc = behind_first_scan(); if (rc != RT_OK) return rc; }
        if (!second_pass_only) {
            hipLaunchKernelGGL(max_abs_hot, dim3(1), dim3(256), 0, s2, sl.d_abs_hot, S, sl.h_abs_hot);
            sl.call.abs_counted = true;
        }
        hipLaunchKernelGGL(plan_pass_b, dim3(S), dim3(256), sizeof(uint32_t) * ((sp.chunks + 31) / 32), s2, sp.full, sp.first,
                           sp.item_chunks, sp.item_count, h->LG, sp.segs_per_chunk, c.n_seg, sp.chunks, sp.blocks_per_stream, h->GPW);
        launch_scan<5>(h, sp, blocks, c.u8, s2);
    } else {
#ifdef RT_STAMPS  // diagnostic build: per-stage cycle sums of every wave of the sparse scan, averaged and printed (stderr)
        static uint32_t *d_dbg = nullptr;
        static size_t dbg_words = 0;
        const size_t words = (size_t)blocks * (scan_block(h->R3) / 64) * kStamps;
        if (words > dbg_words) {
            if (d_dbg) (void)hipFree(d_dbg);
            RT_HIP(h, hipMalloc(&d_dbg, words * sizeof(uint32_t)));
            dbg_words = words;
        }
        RT_HIP(h, hipMemsetAsync(d_dbg, 0, words * sizeof(uint32_t), h->s_scan));
        sp.dbg = d_dbg;
#endif
        launch_scan<0>(h, sp, blocks, c.u8);
#ifdef RT_STAMPS
        {
            std::vector<uint32_t> hd(words);
            RT_HIP(h, hipStreamSynchronize(h->s_scan));
            RT_HIP(h, hipMemcpy(hd.data(), d_dbg, words * sizeof(uint32_t), hipMemcpyDeviceToHost));
            double sum[kStamps] = {};
            double waves = 0;
            for (size_t w = 0; w < words / kStamps; ++w) {
                if (hd[w * kStamps + 11] == 0) continue;
                waves += 1;
                for (int k = 0; k < kStamps; ++k) sum[k] += hd[w * kStamps + k];
            }
            if (const char *dump = RT_DIAG_ENV("RT_STAMPS_DUMP")) {  // raw per-wave words of the last launch, for timelines
                if (FILE *f = std::fopen(dump, "wb")) {
                    std::fwrite(hd.data(), sizeof(uint32_t), words, f);
                    std::fclose(f);
                }
            }
            uint32_t t_min = 0xFFFFFFFFu, t_max = 0;
            double resident = 0;  // wave lifetimes, 100-MHz ticks
            for (size_t w = 0; w < words / kStamps; ++w) {
                if (hd[w * kStamps + 11] == 0) continue;
                t_min = std::min(t_min, hd[w * kStamps + 14]);
                t_max = std::max(t_max, hd[w * kStamps + 15]);
                resident += (double)(hd[w * kStamps + 15] - hd[w * kStamps + 14]);
            }
            const double span = (double)(t_max - t_min);
            std::fprintf(stderr, "RT_STAMPS launch span %.1f us, %.2f waves resident per SIMD on average, %.1f %% of a wave's life inside the step loop; ",
                         span / 100.0, resident / span / 1024.0, 100.0 * sum[13] / resident);
            const double steps = sum[11];
            double tot = 0;
            for (int k = 0; k <= 10; ++k) tot += sum[k];
            std::fprintf(stderr, "RT_STAMPS nperseg %d: %.0f waves, %.1f steps per wave, %.0f cycles per step, in-kernel clock %.0f MHz:", h->N, waves,
                         steps / waves, tot / steps, sum[13] > 0 ? sum[12] / sum[13] * 100.0 : 0.0);
            for (int k = 0; k <= 10; ++k) std::fprintf(stderr, " [%d] %.0f", k, sum[k] / steps);
            std::fprintf(stderr, "\n");
        }
#endif
    }
    RT_HIP(h, hipGetLastError());
    // (ev_scan: the end of the call's scans -- on `sd` where a second scan ran there)
    const bool two_scans = !dense && (mode == RT_MODE_RUNFILTER || mode == RT_MODE_PREFILTER);
    RT_HIP(h, hipEventRecord(sl.ev_scan, two_scans ? s2 : h->s_scan));
    if (sd != (two_scans ? s2 : h->s_scan)) RT_HIP(h, hipStreamWaitEvent(sd, sl.ev_scan, 0));
    DetectArgs a = make_detect_args(h, sl, c.n_seg, h->N, c.n_seg_last);
    a.prev = h->d_tail[c.tail_read];
    a.prev_cols = h->K;
    a.chunks = sp.blocks_per_stream;
    a.spec = h->d_spec;
    a.filtered = (mode == RT_MODE_PREFILTER || mode == RT_MODE_RUNFILTER) ? 1 : 0;
    if (mode == RT_MODE_RUNFILTER) {
        a.seg_total = sp.seg_count + S;  // (the planner's batch total -> pinned host word, by the call's last finalize_records workgroup)
        a.host_seg_total = sl.h_seg_total;
    }
    if (dense) {
        launch_detect_dense(h, S, sd, a);
    } else {
        if (a.work_list) {
            // thousands of streams with a few hundred cells each: a wave per stream (detect_group), the per-list waves only for the
            // streams it leaves -- a looping grid that finds its work list empty as a rule
            hipLaunchKernelGGL(detect_group, dim3((S + 3) / 4), dim3(256), h->lds_small, sd, a);
            hipLaunchKernelGGL(detect_bucket_listed, dim3(std::min(S * kQuarters, 1024)), dim3(256), h->lds_small, sd, a);
        } else {
            const int waves = S * kBuckets;
            hipLaunchKernelGGL(detect_bucket<false>, dim3((waves + 3) / 4), dim3(256), h->lds_small, sd, a);
        }
        // (the large instantiation returns at once for a stream without a bucket over kSmallBucket cells; the per-bucket counters
        // are put back to zero by finalize_records)
        hipLaunchKernelGGL(detect_bucket<true>, dim3(S), dim3(256), h->lds_large, sd, a);  // one workgroup per stream: its 16 buckets in turn
        hipLaunchKernelGGL(finalize_records, dim3(S), dim3(256), 0, sd, a);
    }
    RT_HIP(h, hipGetLastError());
    // no readback: the call's last workgroup wrote the counter words to pinned host memory
    RT_HIP(h, hipEventRecord(sl.ev_done, sd));
    return RT_OK;
}

// AUTO: a few streams of the batch overflowed their candidate lists (one noisy SDR among hundreds): only they are
// analysed again, on the dense path, while the records of the others stand.  The scan and detect_dense take the
// list of streams; the records go behind the ones already in the pool (the counter word is put back first), the
// streams' offset / count entries are simply overwritten.
int enqueue_partial_dense(rt_handle *h, Slot &sl, int n_list, unsigned long long records_so_far) {
    const CallCtx &c = sl.call;
    const int n_part = std::min(h->cfg.n_streams, kMaxPartial);
    if (!h->d_spec_part) {
        const size_t bytes = (size_t)n_part * (size_t)h->max_seg * (size_t)h->N * sizeof(float);
        hipError_t e = hipMalloc(&h->d_spec_part, bytes ? bytes : 4);
        if (e != hipSuccess) {
            h->d_spec_part = nullptr;
            h->err = "dense spectrogram scratch for the partial re-run (" + std::to_string(bytes) + " bytes): " + hipGetErrorString(e);
            return RT_E_NOMEM;
        }
    }
    StftParams sp = make_stft_params(h, sl, c.iq, c.stream_stride, c.n_seg, c.tail_write);
    sp.n_streams = n_list;
    sp.stream_list = sl.h_list;  // pinned host memory, device-visible: a few dozen scalar loads per workgroup
    sp.spec = h->d_spec_part;
    *sl.h_total = records_so_far;
    RT_HIP(h, hipMemcpyAsync(sl.d_counters, sl.h_total, sizeof(unsigned long long), hipMemcpyHostToDevice, h->s_scan));
    launch_scan<1>(h, sp, n_list * sp.blocks_per_stream, c.u8);
    RT_HIP(h, hipGetLastError());
    DetectArgs a = make_detect_args(h, sl, c.n_seg, h->N, c.n_seg_last);
    a.prev = h->d_tail[c.tail_read];
    a.prev_cols = h->K;
    a.chunks = sp.blocks_per_stream;
    a.spec = h->d_spec_part;
    a.stream_list = sl.h_list;
    launch_detect_dense(h, n_list, h->s_scan, a);
    RT_HIP(h, hipGetLastError());
    RT_HIP(h, hipEventRecord(sl.ev_done, h->s_scan));
    return RT_OK;
}

// A call needed more records than the slot's pinned pool holds: a larger pool (the reference appends without limit,
// analyze.py:449-450; here the limit is what the streams' record_capacity can deliver).  Only while none of the slot's
// kernels is in flight.  Failure leaves the old pool in place.
int grow_pool(rt_handle *h, Slot &sl, int64_t want) {
    want = std::min(want, h->pool_max);
    if (want <= sl.pool_cap) return RT_OK;
    int64_t cap = std::min(h->pool_max, std::max(want + want / 4, 2 * sl.pool_cap));
    rt_record *p = nullptr;
    hipError_t e = hipHostMalloc(&p, (size_t)cap * sizeof(rt_record));
    if (e != hipSuccess && cap > want) {
        cap = want;
        e = hipHostMalloc(&p, (size_t)cap * sizeof(rt_record));
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        h->err = "record pool of " + std::to_string(cap) + " records: " + hipGetErrorString(e);
        return RT_E_NOMEM;
    }
    (void)hipHostFree(sl.h_records);
    sl.h_records = p;
    sl.pool_cap = cap;
    h->pool_want = std::max(h->pool_want, cap);
    return RT_OK;
}

// A stream of the call wanted more records than the handle's per-stream capacity (word 4 of the counters): the capacity --
// rt_config.record_capacity is where it STARTS -- grows to hold them, for this call and every later one.  The reference appends
// signals without limit (analyze.py:449-450); here the limit is the memory: per stream and slot 40 bytes a record of raw-record
// area, and what the pinned pool may then be asked to hold.  Everything in flight is waited for first (the other slot's kernels
// write their own raw-record area, which is replaced as well).  Failure leaves the old capacity in place.
int grow_record_capacity(rt_handle *h, unsigned long long wanted) {
    if (wanted <= (unsigned long long)h->rec_cap) return RT_OK;
    if (wanted > (1ull << 24)) wanted = 1ull << 24;  // (a buffer of 2^24 records per stream is beyond any key space the scans have)
    const int cap = next_pow2((int)wanted);
    RT_HIP(h, hipStreamSynchronize(h->s_scan));
    RT_HIP(h, hipStreamSynchronize(h->s_detect));
    const size_t bytes = (size_t)h->cfg.n_streams * (size_t)cap * sizeof(rt_record);
    rt_record *fresh[kSlots] = {nullptr, nullptr};
    for (int i = 0; i < kSlots; ++i) {
        if (hipMalloc(&fresh[i], bytes) != hipSuccess) {
            (void)hipGetLastError();
            for (int j = 0; j < i; ++j) (void)hipFree(fresh[j]);
            h->err = "raw-record area for " + std::to_string(cap) + " records per stream (" + std::to_string(bytes) + " bytes per slot) does not fit the device";
            return RT_E_NOMEM;
        }
    }
    for (int i = 0; i < kSlots; ++i) {
        (void)hipFree(h->slot[i].d_raw);
        h->slot[i].d_raw = fresh[i];
    }
    h->rec_cap = cap;
    h->lds_dense = rec_lds_bytes(cap);
    h->pool_max = ((int64_t)h->cfg.n_streams + std::min(h->cfg.n_streams, kMaxPartial)) * cap;
    return RT_OK;
}

// claim the slot of the next call; the GPU work of the call that used it last must be over
// before its scratch is rewritten (its results, if never fetched, are dropped)
int claim_slot(rt_handle *h, Slot **out, CallCtx *saved) {
    Slot &sl = h->slot[h->n_calls % kSlots];
    // the slot's previous call may still be in its detection (a stream of its own): what is enqueued from here on waits for it
    if (sl.call.seq != 0 && h->s_detect != h->s_scan) (void)hipStreamWaitEvent(h->s_scan, sl.ev_done, 0);
    if (sl.pool_cap < h->pool_want && !sl.call.pending) {
        // the other slot's pool had to grow: this one follows before its next call needs it (best effort)
        if (sl.call.seq == 0 || hipEventSynchronize(sl.ev_done) == hipSuccess) (void)grow_pool(h, sl, h->pool_want);
    }
    *saved = sl.call;  // put back if the new call fails before it has launched anything
    sl.call = CallCtx{};
    sl.call.seq = h->n_calls + 1;
    *out = &sl;
    return RT_OK;
}

// Undo the newest enqueued call of a (lane-less) handle: wait for its kernels, forget it, and put the
// look-back bookkeeping back to what it was before the call.
void rollback_newest(rt_handle *h) {
    Slot *best = nullptr;
    for (auto &sl : h->slot)
        if (sl.call.pending && (!best || sl.call.seq > best->call.seq)) best = &sl;
    if (!best || best->call.seq != h->n_calls) return;
    (void)hipSetDevice(h->cfg.device);
    (void)hipEventSynchronize(best->ev_done);
    const CallCtx &c = best->call;
    if (!c.is_extract) {
        h->tail_cur = c.prev_tail_cur;
        h->n_seg_last = c.prev_n_seg_last;
        h->dense_sticky = c.prev_dense_sticky;
        h->minsum_slot = c.prev_minsum_slot;  // (the chunk minima of the dropped buffer must not set the next call's thresholds)
        if (best->min_seq == c.seq) best->min_items = 0;  // (... nor stand in for the slot's older ones, which they have overwritten)
        if (c.no_last) {
            for (int s = 0; s < h->cfg.n_streams; ++s)
                if (best->h_no_last[s]) h->reset_pending[(size_t)s] = 1;
            h->any_reset_pending = true;
        }
    }
    best->call = CallCtx{};
    h->n_calls--;
}

Slot *oldest_pending(rt_handle *h) {
    Slot *best = nullptr;
    for (auto &sl : h->slot)
        if (sl.call.pending && (!best || sl.call.seq < best->call.seq)) best = &sl;
    return best;
}

// forward one call to every lane; `call(kid, first stream of the kid)`; the first failure is reported.
// `enqueues`: the call enqueues work (rt_process*, rt_extract) -- when lane k fails, the lanes before it
// are rolled back, so that no lane holds a pending call the others lack (FIFO fetches pair calls by position).
template <class F>
int for_each_lane(rt_handle *h, F call, bool enqueues = false) {
    if (enqueues) {
        // The slot every lane is about to claim may still hold a call that was never fetched (a third rt_process without
        // an rt_fetch): it is dropped HERE, in all lanes, before any of them starts -- if a later lane then fails, the
        // lanes before it roll their new call back and all of them are left with the same pending calls (a lane that
        // failed before its first launch used to put the old call back while the others had lost it).
        for (rt_handle *k : h->kids) {
            Slot &sl = k->slot[k->n_calls % kSlots];
            if (sl.call.pending) {
                (void)hipSetDevice(k->cfg.device);
                (void)hipEventSynchronize(sl.ev_done);
                sl.call.pending = false;
            }
        }
    }
    // test hook of the DIAGNOSTIC build (librt_analyze_diag.so; tests/test_gpu_parity.py, fault injection): a handle created under
    // RT_TEST_FAIL_LANE=<k>:<n> has its lane k refuse the handle's n-th enqueue, as a device allocation failure inside that lane
    // would.  The product build never reads the environment (rt_diag.h): test_fail_lane stays -1 there.
    int fail_lane = -1;
#ifdef RT_DIAG
    if (enqueues && h->test_fail_lane >= 0 && ++h->test_enqueues == (uint64_t)h->test_fail_nth) fail_lane = h->test_fail_lane;
#endif
    for (size_t k = 0; k < h->kids.size(); ++k) {
        int rc;
        if ((int)k == fail_lane) {
            h->kids[k]->err = "injected failure";
            rc = RT_E_NOMEM;
        } else {
            rc = call(h->kids[k], (int64_t)h->kid_base[k]);
        }
        if (rc != RT_OK) {
            h->err = h->kids[k]->err;
            if (enqueues)
                for (size_t j = 0; j < k; ++j) rollback_newest(h->kids[j]);
            return rc;
        }
    }
    return RT_OK;
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
    if (!h->kids.empty()) {
        for (rt_handle *k : h->kids) rt_destroy(k);
        delete h;
        return;
    }
    (void)hipSetDevice(h->cfg.device);
    (void)hipDeviceSynchronize();
    (void)hipFree(h->d_work);
    (void)hipFree(h->d_twg);
    (void)hipFree(h->d_cwin);
    (void)hipFree(h->d_bfilt);
    (void)hipFree(h->d_window);
    (void)hipFree(h->d_window_t);
    (void)hipFree(h->d_sub_first);
    (void)hipHostFree(h->h_sub_list);
    (void)hipFree(h->d_tw1);
    (void)hipFree(h->d_tw2);
    for (auto &t : h->d_tail) (void)hipFree(t);
    (void)hipFree(h->d_spec);
    (void)hipFree(h->d_spec_part);
    for (auto &st : h->d_iq_stage) (void)hipFree(st);
    (void)hipFree(h->d_thr_s);
    (void)hipFree(h->d_cal_s);
    for (auto &sl : h->slot) {
        (void)hipFree(sl.d_hot);
        (void)hipFree(sl.d_hot_count);
        (void)hipFree(sl.d_hot_seen);
        (void)hipFree(sl.d_items);
        (void)hipHostFree(sl.h_hot_total);
        (void)hipFree(sl.d_psum);
        (void)hipFree(sl.d_full);
        (void)hipFree(sl.d_cell_hot);
        (void)hipFree(sl.d_cell_need);
        (void)hipFree(sl.d_chunk_min);
        (void)hipFree(sl.d_thr_bin);
        (void)hipFree(sl.d_thr_nat);
        (void)hipFree(sl.d_abs_hot);
        (void)hipHostFree(sl.h_abs_hot);
        (void)hipFree(sl.d_seg_list);
        (void)hipHostFree(sl.h_seg_total);
        (void)hipFree(sl.d_raw);
        (void)hipFree(sl.d_raw_count);
        (void)hipFree(sl.d_counters);
        (void)hipHostFree(sl.h_counters);
        (void)hipHostFree(sl.h_rec_offset);
        (void)hipHostFree(sl.h_rec_count);
        (void)hipHostFree(sl.h_records);
        (void)hipHostFree(sl.h_no_last);
        (void)hipHostFree(sl.h_overflow);
        (void)hipHostFree(sl.h_incons);
        (void)hipHostFree(sl.h_dc_flag);
        (void)hipHostFree(sl.h_list);
        (void)hipHostFree(sl.h_total);
        if (sl.ev_begin) (void)hipEventDestroy(sl.ev_begin);
        if (sl.ev_first) (void)hipEventDestroy(sl.ev_first);
        if (sl.ev_scan) (void)hipEventDestroy(sl.ev_scan);
        if (sl.ev_done) (void)hipEventDestroy(sl.ev_done);
    }
    if (h->own_scan_stream && h->s_detect && h->s_detect != h->s_scan) (void)hipStreamDestroy(h->s_detect);
    if (h->own_scan_stream && h->s_scan) (void)hipStreamDestroy(h->s_scan);
    delete h;
}

int rt_create(const rt_config *cfg, rt_handle **out) {
    if (!cfg || !out) return fail_create(RT_E_INVALID, "null argument");
    *out = nullptr;
#ifdef RT_DIAG
    warn_unknown_switches();
#endif
    if (cfg->n_streams < 1 || cfg->max_samples < 0 || !cfg->window || !(cfg->sample_rate > 0))
        return fail_create(RT_E_INVALID, "n_streams, max_samples, window and sample_rate must be set");
    if (!(cfg->max_duration_s >= 0) || !(cfg->min_duration_s >= 0))
        return fail_create(RT_E_INVALID, "durations must be non-negative");
    int R3 = 0, QS = 0;
    for (int r : {1, 2, 4, 8, 16})
        if (cfg->nperseg == 256 * r) R3 = r;
    for (int q : {2, 4, 8})
        if (cfg->nperseg == 16 * q) {  // 32 / 64 / 128: the fused scans with lane groups of q lanes (rt_kernels.h: stft_scan<.., QS>)
            R3 = 1;
            QS = q;
        }
    int big = 0;
    if (cfg->nperseg == 8192 || cfg->nperseg == 16384) {  // one workgroup per segment (rt_scan_wg.h: stft_wg), sparse and dense path
        R3 = 1;  // (sizes scratch nobody uses at these sizes)
        big = wg_block(cfg->nperseg);
    }
    bool general = false, bluestein = false;
    if (!R3) {
        // every other size the reference may be given (it passes any integer on to SciPy): the other powers of two from 8 to 16 384 by a
        // general LDS transform, everything else from 8 to 8 192 by Bluestein's algorithm on it -- both on the dense path (rt_general.h)
        const int n = cfg->nperseg;
        const bool pow2 = n > 0 && (n & (n - 1)) == 0;
        if (n < 8 || (pow2 && n > kGeneralMaxN) || (!pow2 && n > kGeneralMaxN / 2))
            return fail_create(RT_E_UNSUPPORTED, "fft_nperseg " + std::to_string(n) + " is not supported: 8 ... 8192, or a power of two up to 16384 (the powers of two from 32 "
                                                 "on run the fused scan kernels, every other size a general transform on the dense path)");
        general = true;
        bluestein = !pow2;
        R3 = 1;  // (sizes the scratch the general path does not use)
    }
    if (cfg->mode < RT_MODE_AUTO || cfg->mode > RT_MODE_RUNFILTER) return fail_create(RT_E_INVALID, "bad mode");
    if (general && cfg->mode != RT_MODE_AUTO && cfg->mode != RT_MODE_DENSE)
        return fail_create(RT_E_UNSUPPORTED, "fft_nperseg " + std::to_string(cfg->nperseg) + " runs on the dense path only: mode must be RT_MODE_AUTO or RT_MODE_DENSE");
    if (big && (cfg->mode == RT_MODE_PREFILTER || cfg->mode == RT_MODE_RUNFILTER))
        return fail_create(RT_E_UNSUPPORTED, "fft_nperseg " + std::to_string(cfg->nperseg) + " has the sparse and the dense path only: mode must be RT_MODE_AUTO, RT_MODE_SPARSE or RT_MODE_DENSE");
    if (cfg->lanes > 1 && cfg->n_streams > 1) {
        // stream groups on their own handles and HIP streams: the detection kernels, launch gaps and last
        // workgroup round of one group overlap the scan of another
        if (cfg->hip_stream) return fail_create(RT_E_INVALID, "lanes > 1 run on their own HIP streams: hip_stream must be NULL");
        const int lanes = cfg->lanes < cfg->n_streams ? cfg->lanes : cfg->n_streams;
        rt_handle *p = new (std::nothrow) rt_handle();
        if (!p) return fail_create(RT_E_NOMEM, "out of host memory");
        p->cfg = *cfg;
        if (const char *spec = RT_DIAG_ENV("RT_TEST_FAIL_LANE")) {  // (diagnostic builds only: rt_diag.h)
            int lane = -1, nth = 0;
            if (std::sscanf(spec, "%d:%d", &lane, &nth) == 2 && lane >= 0 && nth > 0) {
                p->test_fail_lane = lane;
                p->test_fail_nth = nth;
            }
        }
        for (int k = 0; k <= lanes; ++k) p->kid_base.push_back((int)((int64_t)cfg->n_streams * k / lanes));
        for (int k = 0; k < lanes; ++k) {
            rt_config kc = *cfg;
            kc.lanes = 1;
            kc.segs_per_chunk = choose_chunk(*cfg, R3, QS, cfg->n_streams, (int)(cfg->max_samples / cfg->nperseg));  // the whole batch's choice
            kc.n_streams = p->kid_base[(size_t)k + 1] - p->kid_base[(size_t)k];
            // (detection by groups of buckets is decided by the whole batch: the lanes' launches run side by side)
            if (cfg->n_streams >= 1024) kc.flags |= (int32_t)kFlagLaneOfLargeBatch;
            rt_handle *kid = nullptr;
            g_creating_lane = true;
            const int rc = rt_create(&kc, &kid);
            g_creating_lane = false;
            if (rc != RT_OK) {
                if (p->kids.empty()) delete p; else rt_destroy(p);
                return rc;  // g_create_error was set by the failing rt_create
            }
            p->kids.push_back(kid);
        }
        *out = p;
        return RT_OK;
    }

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
    h->big = big;
    h->QS = QS;
    h->general = general;
    if (general) {
        h->cfg.mode = RT_MODE_DENSE;
        h->bluestein = bluestein;
        h->gen_m = 1;
        while (h->gen_m < (bluestein ? 2 * cfg->nperseg - 1 : cfg->nperseg)) h->gen_m <<= 1;
        h->log2n = 0;
        while ((1 << h->log2n) < h->gen_m) ++h->log2n;
    }
    h->N = cfg->nperseg;
    h->LG = QS ? QS : 16 * R3;
    h->GPW = big ? 1 : scan_block(R3) / h->LG;  // (stft_wg: a chunk per workgroup)
    h->timing = (cfg->flags & RT_FLAG_TIMING) != 0;
    h->rec_cap = cfg->record_capacity > 0 ? cfg->record_capacity : 1024;
    h->stride = probe_stride(h->N, cfg->sample_rate, cfg->min_duration_s);
    {
        const double hop = seg_time(1, h->N, cfg->sample_rate) - seg_time(0, h->N, cfg->sample_rate);
        const double k = std::floor(cfg->max_duration_s / hop) + 2.0;
        h->K = (int)std::min(k, 1.0e6);
        if (h->K < 1) h->K = 1;
    }
    h->max_seg = (int)(cfg->max_samples / h->N);
    h->L = choose_chunk(*cfg, R3, QS, cfg->n_streams, h->max_seg);  // fixed per handle so the scratch bound holds for every call
    h->max_chunks = std::max(1, (h->max_seg + h->L - 1) / h->L);
    int max_blocks_per_stream = (h->max_chunks + h->GPW - 1) / h->GPW;
    h->max_blocks = max_blocks_per_stream;
    {
        // Run-length pre-filter: a run shorter than r_min cells (and not through t = 0) fails the duration gate whatever
        // else holds -- (len + 1) * hop < signal_min_duration (analyze.py:427-430; rt_core.h: gate_run), with a margin of
        // 1e-9 for the rounding of the float64 expressions.  A run of >= 2 L - 1 cells covers an aligned chunk of L.
        const long long r_min = min_run_cells(h);
        h->prefilter_ok = !general && !big && h->L >= 4 && 2ll * h->L - 1 <= r_min && h->max_seg >= 2 * h->L &&
                          h->max_chunks <= (1 << 18);  // (plan_pass_b keeps a bit per chunk in LDS)
        if (cfg->mode == RT_MODE_PREFILTER && !h->prefilter_ok) {
            delete h;
            return fail_create(RT_E_UNSUPPORTED, "RT_MODE_PREFILTER needs signal_min_duration >= 2 * segs_per_chunk STFT hops");
        }
        // nperseg 4096 without chunk bits (they need chunks of one length): a stream's earliest chunks are half as long, so that
        // the last items a launch hands out are short (rt_kernels.h: chunk_geometry).  A few chunks more than max_seg / L.
        h->two_level = scan_wave64(R3) && !h->prefilter_ok && !RT_DIAG_ENV("RT_EXP_ONE_LEVEL");  // (the variable: A/B runs of diagnostic builds, rt_diag.h)
        if (h->two_level) {
            h->max_chunks += h->max_chunks / 4 + 2;
            max_blocks_per_stream = (h->max_chunks + h->GPW - 1) / h->GPW;
            h->max_blocks = max_blocks_per_stream;
        }
        // Exact run-length pre-filter: any chunk length, any plateau length the planner's counters hold (rt_kernels.h: plan_runs).
        // Built where it is asked for, and in AUTO mode.
        h->run_cells = (int)std::max<long long>(1, std::min<long long>(r_min, 1 << 20));
        // (lane groups of two lanes -- nperseg 32 -- hold half a planner word per row: no exact pre-filter there)
        const bool fits = !general && !big && h->LG >= 4 && std::min<long long>(h->run_cells, (long long)h->max_seg + 1) <= kPlanMaxRun && h->max_seg >= 2;
        if (cfg->mode == RT_MODE_RUNFILTER && !fits) {
            delete h;
            return fail_create(RT_E_UNSUPPORTED, "RT_MODE_RUNFILTER: the minimum plateau length (in STFT hops) is beyond the planner's counters");
        }
        h->runfilter_ok = fits && (cfg->mode == RT_MODE_RUNFILTER || cfg->mode == RT_MODE_AUTO);
        if (h->runfilter_ok && cfg->mode == RT_MODE_AUTO) {
            // In AUTO mode the level is optional -- the only one between the sparse and the dense path at the reference's
            // default geometry, the one above the chunk bits elsewhere (noise far over the absolute threshold: every chunk
            // bit set, while SNR-aware cell bits stay selective): its scratch (two slots of threshold bits + kept cells +
            // segment lists + chunk minima) is taken where it is a small part of what is free -- at most half where it is
            // the only middle level, an eighth where the chunk bits exist -- and AUTO does without it otherwise.
            const size_t per_slot = (size_t)cfg->n_streams * std::max(h->max_seg, 1) * (size_t)(h->LG * 4 + 4) + (size_t)cfg->n_streams * h->N * 4 * (size_t)(h->max_blocks + 2);
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || (size_t)kSlots * per_slot > free_b / (h->prefilter_ok ? 8 : 2)) h->runfilter_ok = false;
        }
    }
    if (!general && ((long long)h->N << key_tbits(std::max(h->max_seg, 2))) > 0x100000000ll) {
        delete h;
        return fail_create(RT_E_UNSUPPORTED, "max_samples too large for 32-bit cell keys");
    }
    // candidate cells per (stream, bucket): by default a full row of one bin fits, times nperseg / 1024 -- a bucket
    // holds nperseg / 16 bins, and with 256 of them (nperseg 4096) a dozen tags per stream put several active
    // bins into one bucket (config 5: up to ~1000 cells per bucket at 781 segments)
    h->hot_cap = general ? 64  // (no candidate lists on the dense path: the smallest scratch)
                 : cfg->hot_capacity > 0
                     ? cfg->hot_capacity
                     : std::min(8192, std::max(kSmallBucket, next_pow2(std::max(h->max_seg, 1)) * std::max(1, h->N / 1024)));
    h->lds_dense = rec_lds_bytes(h->rec_cap);
    {
        // plateaus a wave stages per bucket: a bucket holds N/16 bins; 32 keeps four waves' LDS
        // under 40 KiB (all 16 bucket waves of a CU resident at once) for nperseg 256/512
        h->cand_cap = (h->N / kBuckets <= 32) ? 32 : kCandCapMax;
        const size_t tail = (((size_t)(h->N / kBuckets) * 4 + 15) & ~(size_t)15) + sizeof(rt_record) * h->cand_cap;
        h->lds_large = (size_t)next_pow2(std::max(h->hot_cap, 64)) * 8 + tail;
        h->lds_small = 4 * ((size_t)kSmallBucket * 8 + tail);
    }
    if (h->lds_large + 8 * 1024 > 160 * 1024) {
        delete h;
        return fail_create(RT_E_INVALID, "hot_capacity does not fit the 160 KiB LDS of a CU");
    }
    {
        // Sparse detection by groups of buckets (detect_group) where a launch of one wave per (stream, bucket) would be many rounds of
        // nearly empty waves: from 1 024 streams per handle on (below that the per-bucket waves fit the chip in a round or two and finish
        // sooner than one wave per stream would).  Needs the group's row means beside its cells (nperseg <= 256) and (bin, t) + a
        // 10-bit position in one word.  RT_FLAG_GROUP_DETECT / RT_FLAG_NO_GROUP_DETECT force it either way (tests, A/B runs).
        int fbits = 0;
        while ((1 << fbits) < h->N) ++fbits;
        const bool can = !general && h->N <= kGroupBins && fbits + key_tbits(std::max(h->max_seg, 1)) + 10 <= 32;
        const bool want = (cfg->flags & RT_FLAG_GROUP_DETECT) ? true : (cfg->flags & RT_FLAG_NO_GROUP_DETECT) ? false
                          : cfg->n_streams >= 1024 || (g_creating_lane && (cfg->flags & kFlagLaneOfLargeBatch));
        h->group_detect = can && want;
        h->group_forced = (cfg->flags & RT_FLAG_GROUP_DETECT) != 0;
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
        h->s_scan = static_cast<hipStream_t>(cfg->hip_stream);
        h->s_detect = h->s_scan;  // (a caller's stream: everything in order on it)
    } else {
        RT_CREATE_HIP(hipStreamCreateWithFlags(&h->s_scan, hipStreamNonBlocking));
        h->own_scan_stream = true;
        // A second stream for everything behind a call's first scan (enqueue_analysis: `sd`) -- where it pays.  Measured on one box
        // each, whole path, against everything in order on one stream: config 2 one lane 0.79 -> 0.73 ms per step (profiles/r04_e_*);
        // the exact pre-filter at the reference's default geometry 2.45 -> 2.1 ms (round 5: its planning kernels and second scan run
        // beside the next call's first scan, profiles/r05_b_*).  Not at nperseg >= 1024: behind a persistent grid (a chip-filling
        // launch whose workgroups live until the items run out) the next scan takes the chip first, the tail runs at its very end,
        // and rt_fetch -- and with it the host's next rt_process -- returns a scan later than it could (config 3 14.87 -> 15.7 ms per
        // step; nperseg 4096: 5.09 ms per launch in order against 5.34 with the detection's stream beside it, round 4's csv).
        // The lanes of a laned handle overlap one another's tails already (config 2 two lanes 0.688 -> 0.735 with a second stream
        // per lane).  Stream priorities changed none of this.
        bool second = R3 <= 2 && !big && !g_creating_lane;  // (stft_wg's launches fill the chip many times over: in order, like nperseg >= 1024)
        if (const char *v = RT_DIAG_ENV("RT_EXP_TAIL")) h->tail_mode = (v[0] == 'C') ? 1 : 0;
        if (const char *v = RT_DIAG_ENV("RT_EXP_STREAMS")) second = (v[0] == '2');  // (diagnostic builds: "1" / "2" force the choice, lanes included)
        if (!second) {
            h->s_detect = h->s_scan;
        } else {
            int least = 0, greatest = 0;
            (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
            int prio = greatest;
            if (const char *v = RT_DIAG_ENV("RT_EXP_TAIL_PRIO")) prio = v[0] == 'l' ? least : v[0] == 'n' ? 0 : greatest;
            RT_CREATE_HIP(hipStreamCreateWithPriority(&h->s_detect, hipStreamNonBlocking, prio));
        }
    }

    const int S = cfg->n_streams, N = h->N, LG = h->LG;
    h->reset_pending.assign((size_t)S, 0);
    if (general) {
        const int M = h->gen_m;
        std::vector<cf> twg((size_t)M / 2);
        for (int m = 0; m < M / 2; ++m) {
            const double ang = -6.283185307179586476925286766559 * (double)m / (double)M;
            twg[(size_t)m] = cf{(float)std::cos(ang), (float)std::sin(ang)};
        }
        RT_CREATE_HIP(hipMalloc(&h->d_twg, sizeof(cf) * twg.size()));
        RT_CREATE_HIP(hipMemcpy(h->d_twg, twg.data(), sizeof(cf) * twg.size(), hipMemcpyHostToDevice));
        {
            const void *big_lds[] = {reinterpret_cast<const void *>(stft_general<false>), reinterpret_cast<const void *>(stft_general<true>),
                                     reinterpret_cast<const void *>(stft_bluestein<false, 1, 256>), reinterpret_cast<const void *>(stft_bluestein<true, 1, 256>),
                                     reinterpret_cast<const void *>(stft_bluestein<false, 2, 256>), reinterpret_cast<const void *>(stft_bluestein<true, 2, 256>),
                                     reinterpret_cast<const void *>(stft_bluestein<false, 2, 512>), reinterpret_cast<const void *>(stft_bluestein<true, 2, 512>),
                                     reinterpret_cast<const void *>(stft_bluestein<false, 4, 512>), reinterpret_cast<const void *>(stft_bluestein<true, 4, 512>),
                                     reinterpret_cast<const void *>(stft_bluestein<false, 4, 1024>), reinterpret_cast<const void *>(stft_bluestein<true, 4, 1024>)};
            for (const void *f : big_lds) RT_CREATE_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, padded_len(kGeneralMaxN, 14) * (int)sizeof(cf)));
        }
        if (h->bluestein) {
            // chirp w[n] = exp(-i pi n^2 / N) (the exponent reduced mod 2 N in integers); the window (times sqrt(scale), as every scan
            // takes it) times the chirp; the filter conj(w) on -N < m < N, wrapped to length M, its transform in double precision
            const double pi = 3.14159265358979323846264338327950288;
            auto chirp = [&](long long n, double sign, double *re, double *im) {
                const long long e = (n * n) % (2ll * N);
                const double ang = sign * pi * (double)e / (double)N;
                *re = std::cos(ang);
                *im = std::sin(ang);
            };
            std::vector<cf> cwin((size_t)N);
            const double root = std::sqrt((double)cfg->scale);
            for (int n = 0; n < N; ++n) {
                double cr, ci;
                chirp(n, -1.0, &cr, &ci);
                const double wv = (double)cfg->window[n] * root;
                cwin[(size_t)n] = cf{(float)(wv * cr), (float)(wv * ci)};
            }
            std::vector<double> br((size_t)M, 0.0), bi((size_t)M, 0.0);
            for (int m = 0; m < N; ++m) {
                double cr, ci;
                chirp(m, +1.0, &cr, &ci);
                br[(size_t)m] = cr;
                bi[(size_t)m] = ci;
                if (m) {
                    br[(size_t)(M - m)] = cr;
                    bi[(size_t)(M - m)] = ci;
                }
            }
            {  // in-place radix-2 transform of the filter, float64 (bit reversal, then log2 M stages)
                for (int i = 1, j = 0; i < M; ++i) {
                    int bit = M >> 1;
                    for (; j & bit; bit >>= 1) j ^= bit;
                    j ^= bit;
                    if (i < j) {
                        std::swap(br[(size_t)i], br[(size_t)j]);
                        std::swap(bi[(size_t)i], bi[(size_t)j]);
                    }
                }
                for (int len = 2; len <= M; len <<= 1) {
                    for (int i = 0; i < M; i += len)
                        for (int k = 0; k < len / 2; ++k) {
                            const double ang = -2.0 * pi * (double)k / (double)len;
                            const double wr = std::cos(ang), wi = std::sin(ang);
                            const size_t a = (size_t)(i + k), b = (size_t)(i + k + len / 2);
                            const double xr = br[b] * wr - bi[b] * wi, xi = br[b] * wi + bi[b] * wr;
                            br[b] = br[a] - xr;
                            bi[b] = bi[a] - xi;
                            br[a] += xr;
                            bi[a] += xi;
                        }
                }
            }
            // kept in bit-reversed order: the kernel's first transform (decimation in frequency) leaves its values in that order
            std::vector<cf> bf((size_t)M);
            for (int i = 0; i < M; ++i) {
                unsigned r = 0;
                for (int b = 0; b < h->log2n; ++b) r |= ((unsigned)(i >> b) & 1u) << (h->log2n - 1 - b);
                bf[(size_t)i] = cf{(float)(br[(size_t)r] / M), (float)(bi[(size_t)r] / M)};
            }
            RT_CREATE_HIP(hipMalloc(&h->d_cwin, sizeof(cf) * cwin.size()));
            RT_CREATE_HIP(hipMemcpy(h->d_cwin, cwin.data(), sizeof(cf) * cwin.size(), hipMemcpyHostToDevice));
            RT_CREATE_HIP(hipMalloc(&h->d_bfilt, sizeof(cf) * bf.size()));
            RT_CREATE_HIP(hipMemcpy(h->d_bfilt, bf.data(), sizeof(cf) * bf.size(), hipMemcpyHostToDevice));
        }
    }
    // window and twiddle tables (twiddles in double, rounded once to float32)
    std::vector<cf> tw1((size_t)LG * 16), tw2((size_t)R3 * 16);
    const double two_pi = 6.283185307179586476925286766559;
    for (int a = 0; a < LG; ++a)
        for (int k1 = 0; k1 < 16; ++k1) {
            // (QS: register r = e QS + k1 of lane a holds A[n' = (16 / QS) a + e][k1] and takes W_N^(n' k1))
            const int e = QS ? (((16 / QS) * a + k1 / QS) * (k1 % QS)) % N : (a * k1) % N;
            const double ang = -two_pi * (double)e / (double)N;
            tw1[(size_t)a * 16 + k1] = cf{(float)std::cos(ang), (float)std::sin(ang)};
        }
    if (big) {
        // stft_wg: W_N^(t 2^i), i < 5, as [i][t]; W_BLK^(d p) as [d][p] (rt_scan_wg.h)
        const int R = big / 16;
        tw1.assign((size_t)5 * big, cf{1.f, 0.f});
        for (int i = 0; i < 5; ++i)
            for (int t = 0; t < big; ++t) {
                const double ang = -two_pi * (double)(((long long)t << i) % N) / (double)N;
                tw1[(size_t)i * big + t] = cf{(float)std::cos(ang), (float)std::sin(ang)};
            }
        tw2.assign((size_t)R * 16, cf{1.f, 0.f});
        for (int d = 0; d < R; ++d)
            for (int pp = 0; pp < 16; ++pp) {
                const double ang = -two_pi * (double)((d * pp) % big) / (double)big;
                tw2[(size_t)d * 16 + pp] = cf{(float)std::cos(ang), (float)std::sin(ang)};
            }
    }
    if (scan_wave64(R3)) {
        // stft_scan64: W_N^(ka n1) with n1 = c + 8 d as W^(8 ka d) (rows 0..6, d = 1..7) times W^(ka c) (rows 7..13, c = 1..7), lane ka
        tw1.assign((size_t)kW64TwRows * 64, cf{1.f, 0.f});
        for (int row = 0; row < 14; ++row)
            for (int ka = 0; ka < 64; ++ka) {
                const int e = row < 7 ? 8 * ka * (row + 1) : ka * (row - 6);
                const double ang = -two_pi * (double)(e % N) / (double)N;
                tw1[(size_t)row * 64 + ka] = cf{(float)std::cos(ang), (float)std::sin(ang)};
            }
    }
    for (int b = 0; b < R3 && !big; ++b)
        for (int q1 = 0; q1 < 16; ++q1) {
            // W_LG^(b q1), times the phase W16^(-s q1) that undoes the column rotation s = x1_rotation(b) of exchange 1
            // (rt_kernels.h): together W_LG^((b - s R3) q1), the exponent reduced in integers
            const int s1 = ((16 / R3 - 2) * b) & 15;
            const int e = (((b - s1 * R3) * q1) % LG + LG) % LG;
            const double ang = -two_pi * (double)e / (double)LG;
            tw2[(size_t)b * 16 + q1] = cf{(float)std::cos(ang), (float)std::sin(ang)};
        }
    {
        int cus = 0;
        RT_CREATE_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device));
        if (cus > 0) h->n_cu = cus;
    }
    h->h_sub_first.assign((size_t)cfg->n_streams, 0);
    RT_CREATE_HIP(hipMalloc(&h->d_sub_first, (size_t)cfg->n_streams * sizeof(int32_t)));
    RT_CREATE_HIP(hipMemset(h->d_sub_first, 0, (size_t)cfg->n_streams * sizeof(int32_t)));
    RT_CREATE_HIP(hipHostMalloc(&h->h_sub_list, (size_t)cfg->n_streams * sizeof(int32_t)));
    RT_CREATE_HIP(hipMalloc(&h->d_work, 2 * sizeof(uint32_t)));
    RT_CREATE_HIP(hipMemset(h->d_work, 0, 2 * sizeof(uint32_t)));
    RT_CREATE_HIP(hipMalloc(&h->d_window, sizeof(float) * N));
    RT_CREATE_HIP(hipMalloc(&h->d_tw1, sizeof(cf) * tw1.size()));
    RT_CREATE_HIP(hipMalloc(&h->d_tw2, sizeof(cf) * tw2.size()));
    {
        // |X|^2 * scale is computed as |X'|^2 with X' the transform of the segment under sqrt(scale) * window:
        // one multiplication per output cell less in the scan kernel (each coefficient rounded once, from double)
        std::vector<float> ws((size_t)N);
        const double root = std::sqrt((double)cfg->scale);
        for (int i = 0; i < N; ++i) ws[(size_t)i] = (float)((double)cfg->window[i] * root);
        RT_CREATE_HIP(hipMemcpy(h->d_window, ws.data(), sizeof(float) * N, hipMemcpyHostToDevice));
        if (big) {
            // stft_wg reads the window in thread order: [t][32] = window[t + BLK j]
            std::vector<float> wt((size_t)N);
            for (int t = 0; t < big; ++t)
                for (int j = 0; j < 32; ++j) wt[(size_t)t * 32 + j] = ws[(size_t)t + (size_t)big * j];
            RT_CREATE_HIP(hipMalloc(&h->d_window_t, sizeof(float) * N));
            RT_CREATE_HIP(hipMemcpy(h->d_window_t, wt.data(), sizeof(float) * N, hipMemcpyHostToDevice));
        }
        if (R3 == 16) {
            std::vector<float> wt((size_t)N);
            for (int l = 0; l < LG; ++l)
                for (int m = 0; m < 16; ++m) wt[(size_t)l * 16 + m] = ws[(size_t)l + (size_t)LG * m];
            if (scan_wave64(R3)) {
                // the order stft_scan64's lanes read it: 16-byte pieces [n0][jq][lane] holding the elements m = n0 + 4 (4 jq + e)
                for (int n0 = 0; n0 < 4; ++n0)
                    for (int jq = 0; jq < 4; ++jq)
                        for (int l = 0; l < 64; ++l)
                            for (int e2 = 0; e2 < 4; ++e2)
                                wt[(((size_t)n0 * 4 + jq) * 64 + l) * 4 + e2] = ws[(size_t)l + 64 * (size_t)(n0 + 4 * (4 * jq + e2))];
            }
            RT_CREATE_HIP(hipMalloc(&h->d_window_t, sizeof(float) * N));
            RT_CREATE_HIP(hipMemcpy(h->d_window_t, wt.data(), sizeof(float) * N, hipMemcpyHostToDevice));
        }
        // Transform of the coefficients as the kernel uses them.  If it is real and confined to bins 0 and +-1 (hamming,
        // hann, boxcar: every cosine-sum window of order <= 1 in get_window's periodic form) the constant detrend is
        // applied to the transform (LIN kernels); any other window keeps the subtract-first kernels.
        double wr[3] = {0, 0, 0}, wi[3] = {0, 0, 0};  // W[0], W[1], W[N-1]
        const int ks[3] = {0, 1, N - 1};
        for (int j = 0; j < 3; ++j)
            for (int n = 0; n < N; ++n) {
                const double ang = -two_pi * (double)(((long long)ks[j] * n) % N) / (double)N;
                wr[j] += (double)ws[(size_t)n] * std::cos(ang);
                wi[j] += (double)ws[(size_t)n] * std::sin(ang);
            }
        // the window is of that form iff the three bins reproduce it:  w[n] = (W0 + W1 e^(+i t) + W_(N-1) e^(-i t)) / N
        double wmax = 0.0, dev = 0.0;
        for (int n = 0; n < N; ++n) {
            const double t = two_pi * (double)n / (double)N;
            const double fit = (wr[0] + (wr[1] + wr[2]) * std::cos(t) - (wi[1] - wi[2]) * std::sin(t)) / N;
            wmax = std::max(wmax, std::fabs((double)ws[(size_t)n]));
            dev = std::max(dev, std::fabs((double)ws[(size_t)n] - fit));
        }
        const double w0 = std::fabs(wr[0]);
        bool cosine_sum = w0 > 0.0 && dev <= 1e-6 * wmax;
        for (int j = 0; j < 3; ++j)
            if (std::fabs(wi[j]) > 1e-6 * w0) cosine_sum = false;  // real transform (w[n] = w[N-n])
        const bool ok = cosine_sum && !(cfg->flags & RT_FLAG_NO_LIN_DETREND) && !big;
        h->lin = ok;
        if (ok) {
            for (int j = 0; j < 3; ++j) h->lin_c[j] = (float)(wr[j] / N);
        }
        if (big) {
            // stft_wg subtracts the mean first, in SciPy's order (no linearity form, no guard); what it takes from the fit is the WINDOW:
            // w[n] = c0 + c1 cos(2 pi n / N), c0 = W[0] / N, c1 = (W[1] + W[N-1]) / N (rt_scan_wg.h: WCOS)
            h->wcos = cosine_sum;
            h->lin_c[0] = (float)(wr[0] / N);
            h->lin_c[1] = (float)((wr[1] + wr[2]) / N);
            h->lin_c[2] = 0.f;
        }
    }
    RT_CREATE_HIP(hipMemcpy(h->d_tw1, tw1.data(), sizeof(cf) * tw1.size(), hipMemcpyHostToDevice));
    RT_CREATE_HIP(hipMemcpy(h->d_tw2, tw2.data(), sizeof(cf) * tw2.size(), hipMemcpyHostToDevice));

    const size_t psum_bytes = (size_t)S * max_blocks_per_stream * N * sizeof(float);
    const size_t tail_bytes = (size_t)S * h->K * N * sizeof(float);
    for (auto &t : h->d_tail) RT_CREATE_HIP(hipMalloc(&t, tail_bytes));
    // (the records of streams re-run dense on their own go behind the ones already in the pool: their first lists stay orphaned)
    h->pool_max = ((int64_t)S + std::min(S, kMaxPartial)) * h->rec_cap;
    h->pool_want = std::min<int64_t>(h->pool_max, cfg->record_pool > 0 ? (int64_t)cfg->record_pool : std::min<int64_t>((int64_t)S * h->rec_cap, kInitialPoolRecords));
    for (auto &sl : h->slot) {
        sl.pool_cap = h->pool_want;
        RT_CREATE_HIP(hipMalloc(&sl.d_psum, std::max<size_t>(psum_bytes, 4)));
        if (h->prefilter_ok && cfg->mode != RT_MODE_DENSE)
            RT_CREATE_HIP(hipMalloc(&sl.d_full, (size_t)S * (h->max_chunks + h->L) * LG * sizeof(uint16_t)));  // + the bits of chunk 0 by segment
        if (sl.d_full) RT_CREATE_HIP(hipMalloc(&sl.d_items, ((size_t)S * h->max_blocks * h->GPW + S) * sizeof(int32_t)));
        if (h->runfilter_ok) {
            const size_t cells = (size_t)S * std::max(h->max_seg, 1) * LG;
            RT_CREATE_HIP(hipMalloc(&sl.d_cell_hot, cells * sizeof(uint16_t)));
            RT_CREATE_HIP(hipMalloc(&sl.d_cell_need, cells * sizeof(uint16_t)));
            RT_CREATE_HIP(hipMemset(sl.d_cell_need, 0, cells * sizeof(uint16_t)));  // (all zeros between calls: plan_runs writes only the words that keep anything)
            RT_CREATE_HIP(hipMalloc(&sl.d_chunk_min, std::max<size_t>(psum_bytes, 4)));  // (a row per work item, like psum)
            RT_CREATE_HIP(hipMalloc(&sl.d_thr_bin, (size_t)S * N * sizeof(float)));
            RT_CREATE_HIP(hipMalloc(&sl.d_thr_nat, (size_t)S * N * sizeof(float)));
            RT_CREATE_HIP(hipMalloc(&sl.d_seg_list, ((size_t)S * std::max(h->max_seg, 1) + S + 1) * sizeof(int32_t)));
            RT_CREATE_HIP(hipHostMalloc(&sl.h_seg_total, sizeof(int32_t)));
            *sl.h_seg_total = 0;
        }
        if (sl.d_full || sl.d_cell_hot) {
            RT_CREATE_HIP(hipMalloc(&sl.d_abs_hot, (size_t)S * sizeof(uint32_t)));
            RT_CREATE_HIP(hipMemset(sl.d_abs_hot, 0, (size_t)S * sizeof(uint32_t)));
            RT_CREATE_HIP(hipHostMalloc(&sl.h_abs_hot, sizeof(uint32_t)));
            *sl.h_abs_hot = 0u;
        }
        RT_CREATE_HIP(hipMalloc(&sl.d_hot, (size_t)S * kBuckets * h->hot_cap * sizeof(uint2)));
        RT_CREATE_HIP(hipMalloc(&sl.d_hot_count, (size_t)S * kBuckets * sizeof(uint32_t)));
        RT_CREATE_HIP(hipMalloc(&sl.d_hot_seen, (size_t)S * kBuckets * sizeof(uint32_t)));
        RT_CREATE_HIP(hipMemset(sl.d_hot_seen, 0, (size_t)S * kBuckets * sizeof(uint32_t)));
        RT_CREATE_HIP(hipHostMalloc(&sl.h_hot_total, (size_t)S * sizeof(int32_t)));
        std::memset(sl.h_hot_total, 0, (size_t)S * sizeof(int32_t));
        RT_CREATE_HIP(hipMalloc(&sl.d_raw, (size_t)S * h->rec_cap * sizeof(rt_record)));
        RT_CREATE_HIP(hipMalloc(&sl.d_raw_count, (size_t)S * sizeof(int32_t)));
        RT_CREATE_HIP(hipMalloc(&sl.d_counters, kCounterWords * sizeof(unsigned long long)));
        RT_CREATE_HIP(hipMemset(sl.d_hot_count, 0, (size_t)S * kBuckets * sizeof(uint32_t)));
        RT_CREATE_HIP(hipMemset(sl.d_raw_count, 0, (size_t)S * sizeof(int32_t)));
        RT_CREATE_HIP(hipMemset(sl.d_counters, 0, kCounterWords * sizeof(unsigned long long)));
        RT_CREATE_HIP(hipHostMalloc(&sl.h_counters, kCounterWords * sizeof(unsigned long long)));
        RT_CREATE_HIP(hipHostMalloc(&sl.h_rec_offset, (size_t)S * sizeof(int32_t)));
        RT_CREATE_HIP(hipHostMalloc(&sl.h_rec_count, (size_t)S * sizeof(int32_t)));
        RT_CREATE_HIP(hipHostMalloc(&sl.h_records, (size_t)sl.pool_cap * sizeof(rt_record)));
        RT_CREATE_HIP(hipHostMalloc(&sl.h_no_last, (size_t)S * sizeof(int32_t)));
        std::memset(sl.h_no_last, 0, (size_t)S * sizeof(int32_t));
        RT_CREATE_HIP(hipHostMalloc(&sl.h_overflow, (size_t)S * sizeof(int32_t)));
        std::memset(sl.h_overflow, 0, (size_t)S * sizeof(int32_t));
        RT_CREATE_HIP(hipHostMalloc(&sl.h_incons, (size_t)S * sizeof(int32_t)));
        std::memset(sl.h_incons, 0, (size_t)S * sizeof(int32_t));
        RT_CREATE_HIP(hipHostMalloc(&sl.h_dc_flag, (size_t)S * sizeof(int32_t)));
        std::memset(sl.h_dc_flag, 0, (size_t)S * sizeof(int32_t));
        RT_CREATE_HIP(hipHostMalloc(&sl.h_list, (size_t)kMaxPartial * sizeof(int32_t)));
        RT_CREATE_HIP(hipHostMalloc(&sl.h_total, sizeof(unsigned long long)));
        RT_CREATE_HIP(hipEventCreate(&sl.ev_begin));
        RT_CREATE_HIP(hipEventCreateWithFlags(&sl.ev_first, hipEventDisableTiming));
        RT_CREATE_HIP(hipEventCreate(&sl.ev_scan));
        RT_CREATE_HIP(hipEventCreate(&sl.ev_done));
    }

    if (big) {
#define RT_WG_SET(B_, M_, U_, W_) RT_CREATE_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(stft_wg<B_, M_, U_, W_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wg_lds_bytes(B_)))
#define RT_WG_SET4(B_, M_) RT_WG_SET(B_, M_, false, false); RT_WG_SET(B_, M_, false, true); RT_WG_SET(B_, M_, true, false); RT_WG_SET(B_, M_, true, true)
        RT_WG_SET4(256, 0); RT_WG_SET4(256, 1); RT_WG_SET4(256, 2);
        RT_WG_SET4(512, 0); RT_WG_SET4(512, 1); RT_WG_SET4(512, 2);
#undef RT_WG_SET4
#undef RT_WG_SET
    }
    RT_CREATE_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(detect_bucket<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_large));
    RT_CREATE_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(detect_bucket<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_small));
    RT_CREATE_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(detect_group), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_small));
    RT_CREATE_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(detect_bucket_listed), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_small));
    RT_CREATE_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(detect_dense<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)rec_lds_bytes(kDenseLdsRecords)));
#undef RT_CREATE_HIP
    *out = h;
    return RT_OK;
}

int rt_reset(rt_handle *h) {
    if (!h) return RT_E_INVALID;
    for (rt_handle *k : h->kids) rt_reset(k);
    h->n_seg_last = -1;
    std::fill(h->reset_pending.begin(), h->reset_pending.end(), (uint8_t)0);
    h->any_reset_pending = false;
    return RT_OK;
}

int rt_reset_stream(rt_handle *h, int32_t stream) {
    if (!h) return RT_E_INVALID;
    if (stream < 0 || stream >= h->cfg.n_streams) {
        h->err = "stream index out of range";
        return RT_E_INVALID;
    }
    if (!h->kids.empty()) {
        for (size_t k = 0; k < h->kids.size(); ++k)
            if (stream < h->kid_base[k + 1]) return rt_reset_stream(h->kids[k], stream - h->kid_base[k]);
        return RT_E_INVALID;
    }
    h->reset_pending[(size_t)stream] = 1;
    h->any_reset_pending = true;
    return RT_OK;
}

int rt_set_stream_params(rt_handle *h, const float *threshold, const float *calibration_db) {
    if (!h) return RT_E_INVALID;
    if (!h->kids.empty())
        return for_each_lane(h, [&](rt_handle *k, int64_t s0) {
            return rt_set_stream_params(k, threshold ? threshold + s0 : nullptr, calibration_db ? calibration_db + s0 : nullptr);
        });
    if (oldest_pending(h)) {
        // a pending call may still be re-run when it is fetched (AUTO mode): it must see the thresholds it was enqueued with
        h->err = "rt_set_stream_params with unfetched calls pending: fetch them first";
        return RT_E_INVALID;
    }
    RT_HIP(h, hipSetDevice(h->cfg.device));
    // kernels in flight read the arrays: let them finish first (a configuration call, not on the hot path)
    RT_HIP(h, hipStreamSynchronize(h->s_scan));
    RT_HIP(h, hipStreamSynchronize(h->s_detect));
    const size_t bytes = (size_t)h->cfg.n_streams * sizeof(float);
    auto put = [&](float *&dst, const float *src) -> int {
        if (!src) {
            if (dst) (void)hipFree(dst);
            dst = nullptr;
            return RT_OK;
        }
        if (!dst) RT_HIP(h, hipMalloc(&dst, bytes));
        RT_HIP(h, hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
        return RT_OK;
    };
    // A stream whose threshold changes starts without look-back: the reference fixes the threshold when a SignalAnalyzer is
    // built (analyze.py:115), a new one needs a new analyzer (_spectrogram_last = None) -- and the sparse scans keep only
    // those tail cells a walk with the threshold of their own call can reach (rt_kernels.h: sparse tail).
    for (int s = 0; s < h->cfg.n_streams; ++s) {
        const float was = h->h_thr_s.empty() ? h->cfg.threshold : h->h_thr_s[(size_t)s];
        const float is = threshold ? threshold[s] : h->cfg.threshold;
        if (!(was == is)) {
            h->reset_pending[(size_t)s] = 1;
            h->any_reset_pending = true;
        }
    }
    if (threshold) h->h_thr_s.assign(threshold, threshold + h->cfg.n_streams); else h->h_thr_s.clear();
    int rc = put(h->d_thr_s, threshold);
    if (rc != RT_OK) return rc;
    return put(h->d_cal_s, calibration_db);
}


static int process_impl(rt_handle *h, const void *iq_dev, int64_t n_samples, int64_t stream_stride, bool u8);

int rt_process(rt_handle *h, const void *iq_dev, int64_t n_samples, int64_t stream_stride) {
    return process_impl(h, iq_dev, n_samples, stream_stride, false);
}

int rt_process_u8(rt_handle *h, const void *iq_u8_dev, int64_t n_samples, int64_t stream_stride) {
    return process_impl(h, iq_u8_dev, n_samples, stream_stride, true);
}

static int process_impl(rt_handle *h, const void *iq_dev, int64_t n_samples, int64_t stream_stride, bool u8) {
    if (!h) return RT_E_INVALID;
    if (!h->kids.empty()) {
        const int64_t bytes = u8 ? 2 : (int64_t)sizeof(cf);
        return for_each_lane(h, [&](rt_handle *k, int64_t s0) {
            const char *base = iq_dev ? static_cast<const char *>(iq_dev) + s0 * stream_stride * bytes : nullptr;
            return process_impl(k, base, n_samples, stream_stride, u8);
        }, true);
    }
    if (!iq_dev && n_samples > 0) {
        h->err = "null IQ pointer";
        return RT_E_INVALID;
    }
    if (n_samples < 0 || n_samples > h->cfg.max_samples || stream_stride < n_samples) {
        h->err = "n_samples/stream_stride out of range for this handle";
        return RT_E_INVALID;
    }
    // the scan kernel loads whole samples (8-byte complex64 / 2-byte I,Q pairs): a misaligned pointer would
    // fault on the device, so it is refused here
    if (reinterpret_cast<uintptr_t>(iq_dev) % (u8 ? 2u : 8u) != 0) {
        h->err = u8 ? "IQ pointer must be 2-byte aligned (uint8 I,Q pairs)" : "IQ pointer must be 8-byte aligned (complex64)";
        return RT_E_INVALID;
    }
    RT_HIP(h, hipSetDevice(h->cfg.device));
    const int T = (int)(n_samples / h->N);
    if (T == 1) {
        h->err = "exactly one segment: the reference raises IndexError (times[1])";
        return RT_E_ONE_SEGMENT;
    }
    Slot *slp = nullptr;
    CallCtx saved;
    int rc = claim_slot(h, &slp, &saved);
    if (rc != RT_OK) return rc;
    Slot &sl = *slp;
    CallCtx &c = sl.call;
    c.prev_tail_cur = h->tail_cur;
    c.prev_n_seg_last = h->n_seg_last;
    c.prev_dense_sticky = h->dense_sticky;
    c.prev_minsum_slot = h->minsum_slot;
    c.iq = iq_dev;
    c.u8 = u8;
    c.n_samples = n_samples;
    c.stream_stride = stream_stride;
    c.n_seg = T;
    c.tail_read = h->tail_cur;
    c.tail_write = (h->tail_cur + 1) % kTails;
    c.n_seg_last = h->n_seg_last;
    // Everything that can fail from here on runs inside `enqueue`; one block below undoes a failed call: nothing stays
    // enqueued, the look-back state and the pending stream resets are the ones before the call, and the call the slot
    // held (possibly still unfetched) is put back unless a launch has already rewritten its scratch (include/rt_analyze.h).
    bool launched = false;
    auto enqueue = [&]() -> int {
        if (h->any_reset_pending) {
            // the slot's previous call (two calls back) may still be reading its flags if it was never fetched
            if (sl.ev_done && h->n_calls >= (uint64_t)kSlots) RT_HIP(h, hipEventSynchronize(sl.ev_done));
            for (int s = 0; s < h->cfg.n_streams; ++s) sl.h_no_last[s] = h->reset_pending[(size_t)s];
            std::fill(h->reset_pending.begin(), h->reset_pending.end(), (uint8_t)0);
            h->any_reset_pending = false;
            c.no_last = true;
        }
        c.mode_used = h->cfg.mode;
        if (h->cfg.mode == RT_MODE_AUTO) {
            // the level the input last needed, for `dense_sticky` more calls; then a probe of the level below
            if (h->dense_sticky > 0) {
                c.mode_used = h->auto_level;
                --h->dense_sticky;
            } else {
                c.mode_used = level_down(h, h->auto_level);
                // (no probe, no back-off where the last count rules the level below out -- rt_core.h: probe_ruled_out; the next
                // call looks again)
                if (probe_ruled_out(c.mode_used, h->auto_level, h->abs_hot_valid, h->abs_hot_seen, (uint64_t)kBuckets * (uint64_t)h->hot_cap,
                                    (uint64_t)T * (uint64_t)h->N))
                    c.mode_used = h->auto_level;
                // one probe at a time: the calls enqueued before this one's verdict is in (the caller may keep a slot's
                // worth of calls in flight) stay on the handle's level instead of each paying for a failed probe
                if (c.mode_used != h->auto_level) h->dense_sticky = kSlots;
            }
        }
        if (T == 0) {
            // empty spectrogram: no signals; `_spectrogram_last` becomes an empty map
            const size_t sb = (size_t)h->cfg.n_streams * sizeof(int32_t);
            launched = true;  // (the slot's result arrays are rewritten from here on)
            RT_HIP(h, hipMemsetAsync(sl.d_counters, 0, kCounterWords * sizeof(unsigned long long), h->s_scan));
            RT_HIP(h, hipMemsetAsync(sl.h_rec_count, 0, sb, h->s_scan));
            RT_HIP(h, hipMemsetAsync(sl.h_rec_offset, 0, sb, h->s_scan));
            RT_HIP(h, hipEventRecord(sl.ev_begin, h->s_scan));
            RT_HIP(h, hipEventRecord(sl.ev_scan, h->s_scan));
            const int rc2 = enqueue_readback(h, sl, h->s_scan);
            if (rc2 != RT_OK) return rc2;
            RT_HIP(h, hipEventRecord(sl.ev_done, h->s_scan));
            return RT_OK;
        }
        return enqueue_analysis(h, sl, c.mode_used, &launched);
    };
    rc = enqueue();
    if (rc != RT_OK) {
        h->dense_sticky = c.prev_dense_sticky;
        h->minsum_slot = c.prev_minsum_slot;
        if (c.no_last) {
            for (int s = 0; s < h->cfg.n_streams; ++s)
                if (sl.h_no_last[s]) h->reset_pending[(size_t)s] = 1;
            h->any_reset_pending = true;
        }
        if (launched) {
            (void)hipStreamSynchronize(h->s_scan);
            (void)hipStreamSynchronize(h->s_detect);
            sl.call = CallCtx{};
        } else {
            sl.call = saved;
        }
        return rc;
    }
    c.pending = true;
    h->n_calls++;
    h->tail_cur = c.tail_write;
    h->n_seg_last = T;
    return RT_OK;
}

static int process_host_impl(rt_handle *h, const void *iq_host, int64_t n_samples, int64_t stream_stride, bool u8) {
    if (!h) return RT_E_INVALID;
    const size_t sample_bytes = u8 ? 2 : sizeof(cf);
    if (!h->kids.empty())
        return for_each_lane(h, [&](rt_handle *k, int64_t s0) {
            const char *base = iq_host ? static_cast<const char *>(iq_host) + s0 * stream_stride * (int64_t)sample_bytes : nullptr;
            return process_host_impl(k, base, n_samples, stream_stride, u8);
        }, true);
    if (n_samples < 0 || n_samples > h->cfg.max_samples || stream_stride < n_samples || (!iq_host && n_samples > 0)) {
        h->err = "n_samples/stream_stride out of range for this handle, or null IQ pointer";
        return RT_E_INVALID;
    }
    RT_HIP(h, hipSetDevice(h->cfg.device));
    const size_t bytes = (size_t)h->cfg.n_streams * (size_t)stream_stride * sample_bytes;
    // One staging buffer per call slot.  The call that used this slot two calls ago is over or dropped by now
    // (at most two are in flight); the one still in flight keeps its own buffer -- it may be re-run from it
    // when it is fetched (AUTO mode, candidate overflow), so it must not be overwritten by this call.
    const int which = (int)(h->n_calls % kSlots);
    Slot &prev = h->slot[which];
    if (prev.call.seq) RT_HIP(h, hipEventSynchronize(prev.ev_done));
    if (bytes > h->iq_stage_bytes[which]) {
        if (h->d_iq_stage[which]) (void)hipFree(h->d_iq_stage[which]);
        h->d_iq_stage[which] = nullptr;
        h->iq_stage_bytes[which] = 0;
        hipError_t e = hipMalloc(&h->d_iq_stage[which], bytes);
        if (e != hipSuccess) {
            h->err = std::string("IQ staging buffer: ") + hipGetErrorString(e);
            return RT_E_NOMEM;
        }
        h->iq_stage_bytes[which] = bytes;
    }
    // blocking copy: the caller may reuse or free its (pageable) buffer as soon as this returns
    if (bytes) RT_HIP(h, hipMemcpy(h->d_iq_stage[which], iq_host, bytes, hipMemcpyHostToDevice));
    return process_impl(h, h->d_iq_stage[which], n_samples, stream_stride, u8);
}

int rt_process_host(rt_handle *h, const void *iq_host, int64_t n_samples, int64_t stream_stride) {
    return process_host_impl(h, iq_host, n_samples, stream_stride, false);
}

int rt_process_u8_host(rt_handle *h, const void *iq_u8_host, int64_t n_samples, int64_t stream_stride) {
    return process_host_impl(h, iq_u8_host, n_samples, stream_stride, true);
}

int rt_extract(rt_handle *h, const float *spec_dev, int32_t n_seg, int32_t n_bins, const float *last_dev,
               int32_t n_seg_last) {
    if (!h) return RT_E_INVALID;
    if (!h->kids.empty())
        return for_each_lane(h, [&](rt_handle *k, int64_t s0) {
            const float *sp = spec_dev ? spec_dev + s0 * n_seg * n_bins : nullptr;
            const float *la = last_dev ? last_dev + s0 * n_seg_last * n_bins : nullptr;
            return rt_extract(k, sp, n_seg, n_bins, la, n_seg_last);
        }, true);
    if (n_seg < 0 || n_bins < 1 || (n_seg > 0 && !spec_dev) || (last_dev && n_seg_last < 0)) {
        h->err = "bad spectrogram arguments";
        return RT_E_INVALID;
    }
    if (n_seg == 1) {
        h->err = "exactly one segment: the reference raises IndexError (times[1])";
        return RT_E_ONE_SEGMENT;
    }
    if (reinterpret_cast<uintptr_t>(spec_dev) % 4u != 0 || reinterpret_cast<uintptr_t>(last_dev) % 4u != 0) {
        h->err = "spectrogram pointers must be 4-byte aligned (float32)";
        return RT_E_INVALID;
    }
    if ((int64_t)n_seg * n_bins > 0x7FFFFFFFll) {
        h->err = "spectrogram too large";
        return RT_E_UNSUPPORTED;
    }
    RT_HIP(h, hipSetDevice(h->cfg.device));
    Slot *slp = nullptr;
    CallCtx saved;
    int rc = claim_slot(h, &slp, &saved);
    if (rc != RT_OK) return rc;
    Slot &sl = *slp;
    CallCtx &c = sl.call;
    c.n_seg = n_seg;
    c.is_extract = true;
    c.mode_used = RT_MODE_DENSE;
    const size_t sb = (size_t)h->cfg.n_streams * sizeof(int32_t);
    RT_HIP(h, hipMemsetAsync(sl.d_counters, 0, kCounterWords * sizeof(unsigned long long), h->s_scan));
    RT_HIP(h, hipEventRecord(sl.ev_begin, h->s_scan));
    RT_HIP(h, hipEventRecord(sl.ev_scan, h->s_scan));
    if (n_seg == 0) {
        RT_HIP(h, hipMemsetAsync(sl.h_rec_count, 0, sb, h->s_scan));
        RT_HIP(h, hipMemsetAsync(sl.h_rec_offset, 0, sb, h->s_scan));
    } else {
        DetectArgs a = make_detect_args(h, sl, n_seg, n_bins, last_dev ? n_seg_last : -1);
        a.dp.tail_cols = last_dev ? n_seg_last : 0;
        a.prev = last_dev;
        a.prev_cols = last_dev ? n_seg_last : 0;
        a.spec = spec_dev;
        a.psum = nullptr;  // caller-supplied map: the kernel sums the rows itself
        launch_detect_dense(h, h->cfg.n_streams, h->s_scan, a);
        RT_HIP(h, hipGetLastError());
    }
    if (n_seg == 0) {
        rc = enqueue_readback(h, sl, h->s_scan);
        if (rc != RT_OK) return rc;
    }
    RT_HIP(h, hipEventRecord(sl.ev_done, h->s_scan));
    c.pending = true;
    h->n_calls++;
    return RT_OK;
}

// peek-mode results of fetch_one: the call produced no usable result (the caller decides what happens to it)
constexpr int kCallFailed = -100;          // -> RT_E_HOT_OVERFLOW (candidate lists overflowed in sparse mode)
constexpr int kCallFailedInternal = -101;  // -> RT_E_HIP

// drop the oldest pending call of a handle (its GPU work is waited for first)
static void discard_oldest(rt_handle *h) {
    Slot *sl = oldest_pending(h);
    if (!sl) return;
    (void)hipSetDevice(h->cfg.device);
    (void)hipEventSynchronize(sl->ev_done);
    sl->call.pending = false;
}

// A call is about to be analysed again from rt_fetch (level-up, stale thresholds, pool growth, the detrend guard) while a later
// call may be in flight: its re-run scan rewrites the look-back columns that later call's detection reads on the handle's second
// stream (the same values, or a superset of the cells -- still an unordered read / write pair), so the re-run waits for it.
static int before_rerun(rt_handle *h, Slot &sl) {
    if (h->s_detect != h->s_scan)
        for (auto &o : h->slot)
            if (&o != &sl && o.call.pending) RT_HIP(h, hipStreamWaitEvent(h->s_scan, o.ev_done, 0));
    return RT_OK;
}

// rt_fetch of one (lane-less) handle.  `peek`: wait, settle fall-backs and count, but deliver nothing and
// keep the call pending even when it has no records (the laned rt_fetch sizes all lanes before it copies).
static int fetch_one(rt_handle *h, rt_record *out, size_t cap, size_t *n_out, bool peek) {
    *n_out = 0;
    Slot *slp = oldest_pending(h);
    if (!slp) {
        h->err = "rt_fetch without a preceding successful rt_process/rt_extract";
        return RT_E_INVALID;
    }
    Slot &sl = *slp;
    CallCtx &c = sl.call;
    RT_HIP(h, hipSetDevice(h->cfg.device));
    RT_HIP(h, hipEventSynchronize(sl.ev_done));
    unsigned long long flags = sl.h_counters[2];
    h->info = rt_call_info{};
    h->info.n_seg = c.n_seg;
    h->info.segs_per_chunk = h->L;
    h->info.n_hot = 0;
    // Guard of the detrend by linearity (rt_kernels.h: StftParams::dc_flag).  The form carries a stream's constant offset through
    // the transform: harmless while the offset is <= 60 dB over the per-sample noise (any <= 16-bit front end; <= 0.02 dB against
    // the oracle), 0.06 - 0.17 dB at 80 dB.  A scan that meets such a stream marks it; from then on that stream -- and only that
    // stream -- is detrended in SciPy's order (_spectral_py.py:2191-2194), and every call whose kernels were enqueued before the
    // mark (this one, and any in flight) is analysed again.
    if (c.ran_lin && !c.is_extract && c.n_seg > 0) {
        bool newly = false;
        for (int s = 0; s < h->cfg.n_streams; ++s) {
            if (sl.h_dc_flag[s] != 0 && !h->h_sub_first[(size_t)s]) {
                h->h_sub_first[(size_t)s] = 1;
                newly = true;
            }
            sl.h_dc_flag[s] = 0;
        }
        if (newly) {
            // (both of the handle's streams drained first, so that no kernel in flight reads the list while it changes)
            RT_HIP(h, hipStreamSynchronize(h->s_scan));
            RT_HIP(h, hipStreamSynchronize(h->s_detect));
            h->n_sub = 0;
            for (int s = 0; s < h->cfg.n_streams; ++s)
                if (h->h_sub_first[(size_t)s]) h->h_sub_list[h->n_sub++] = s;
            RT_HIP(h, hipMemcpy(h->d_sub_first, h->h_sub_first.data(), h->h_sub_first.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            ++h->sub_epoch;
        }
        if (c.sub_epoch != h->sub_epoch) {
            for (int s = 0; s < h->cfg.n_streams; ++s) sl.h_overflow[s] = sl.h_incons[s] = 0;
            c.n_dense_streams = 0;
            int rc = before_rerun(h, sl);
            if (rc == RT_OK) rc = enqueue_analysis(h, sl, c.mode_used);  // (takes the current epoch)
            if (rc != RT_OK) return rc;
            // The call enqueued behind this one (at most one is in flight) was analysed under the old set as well, and the next
            // rt_process will read the look-back columns it wrote: it is analysed again HERE, in sequence order, before anything
            // newer can be enqueued -- not at its own fetch, by which time a newer call's detection would have read the old
            // columns or raced the rewrite (advisor, round 4).  Once per marking.
            for (auto &o : h->slot) {
                if (&o == &sl || !o.call.pending || o.call.seq < c.seq || !o.call.ran_lin || o.call.is_extract || o.call.n_seg <= 0 ||
                    o.call.sub_epoch == h->sub_epoch)
                    continue;
                RT_HIP(h, hipEventSynchronize(o.ev_done));
                for (int s = 0; s < h->cfg.n_streams; ++s) o.h_overflow[s] = o.h_incons[s] = o.h_dc_flag[s] = 0;
                rc = enqueue_analysis(h, o, o.call.mode_used);  // (behind this call's re-run on s_scan; its scratch is its slot's)
                if (rc != RT_OK) return rc;
            }
            RT_HIP(h, hipEventSynchronize(sl.ev_done));
            flags = sl.h_counters[2];
            for (int s = 0; s < h->cfg.n_streams; ++s) sl.h_dc_flag[s] = 0;  // (marked streams mark themselves again: nothing new)
        }
    }
  for (;;) {  // (a second round only after the record pool had to grow)
    RT_TRACE_FETCH(h, sl, "round");
    while ((flags & kFlagHotOverflow) && !c.is_extract && c.mode_used != RT_MODE_DENSE) {
        RT_TRACE_FETCH(h, sl, "hot overflow");
        // which streams overflowed?  (the flags are consumed here, whatever happens next)  The scan stops emitting for a
        // stream once one of its lists has overflowed, so a "run without its preceding cell" in such a stream is not an
        // internal error; in any other stream it is.
        int n_bad = 0;
        bool incons_elsewhere = false;
        for (int s = 0; s < h->cfg.n_streams; ++s) {
            if (sl.h_incons[s] && !sl.h_overflow[s]) incons_elsewhere = true;
            sl.h_incons[s] = 0;
            if (sl.h_overflow[s]) {
                if (n_bad < kMaxPartial) sl.h_list[n_bad] = s;
                ++n_bad;
                sl.h_overflow[s] = 0;
            }
        }
        // The per-bin thresholds of the exact pre-filter were too high for some streams (their noise floor fell from the
        // buffer before to this one): not a matter of capacity.  A few streams of many go dense on their own below (AUTO);
        // otherwise the call is analysed again on the same level, its thresholds now taken from its own row means --
        // the handle's level does not change, and an explicit RT_MODE_RUNFILTER handle does not fail.
        const bool stale = (flags & kFlagThrStale) && c.mode_used == RT_MODE_RUNFILTER && !c.thr_rerun;
        flags &= ~kFlagThrStale;
        const bool few = n_bad > 0 && n_bad <= kMaxPartial && 4 * n_bad <= h->cfg.n_streams &&
                         !(c.mode_used == RT_MODE_SPARSE && level_up(h, RT_MODE_SPARSE) != RT_MODE_DENSE);
        if (stale && !(h->cfg.mode == RT_MODE_AUTO && few)) {
            c.thr_rerun = true;
            int rc = before_rerun(h, sl);
            if (rc == RT_OK) rc = enqueue_analysis(h, sl, RT_MODE_RUNFILTER, nullptr, false, true);
            if (rc != RT_OK) return rc;
            RT_HIP(h, hipEventSynchronize(sl.ev_done));
            flags = sl.h_counters[2];
            continue;
        }
        if (h->cfg.mode != RT_MODE_AUTO) {
            h->err = "candidate-cell capacity exceeded (hot_capacity)";
            if (peek) return kCallFailed;  // the laned rt_fetch drops this call in every lane together
            c.pending = false;
            return RT_E_HOT_OVERFLOW;
        }
        {
            // a few of many: only they go dense, the handle stays on its level
            // (where the pre-filter level is still ahead, the whole batch goes there first: its second pass costs less
            // than the fixed ~1 ms of a dense re-run of a few streams -- one workgroup per stream in detect_dense)
            if (few) {
                const unsigned long long other = flags & ~(kFlagHotOverflow | (incons_elsewhere ? 0ull : kFlagInconsistent));
                const unsigned long long wanted_so_far = sl.h_counters[4];  // (a stream OUTSIDE the few may have outgrown its record capacity)
                int rc = before_rerun(h, sl);
                if (rc == RT_OK) rc = enqueue_partial_dense(h, sl, n_bad, sl.h_counters[0]);
                if (rc == RT_OK) {
                    RT_HIP(h, hipEventSynchronize(sl.ev_done));
                    flags = other | sl.h_counters[2];
                    sl.h_counters[2] = flags;  // (the laned rt_fetch looks at this call twice: sizing, then delivery)
                    // ... and what that stream wanted survives the partial run's own counter words: the record-overflow flag kept in `other`
                    // without it was a truncated delivery (round 6's soak, seed 64 case 56)
                    sl.h_counters[4] = std::max(sl.h_counters[4], wanted_so_far);
                    c.fell_back = true;
                    c.n_dense_streams = n_bad;
                    break;
                }
                // the partial re-run could not be enqueued (its scratch spectrogram did not fit): the whole batch goes one
                // level up instead, which needs no scratch of its own or reports its own failure
                if (rc != RT_E_NOMEM) return rc;
            }
        }
        // Re-run of the same buffer with the same look-back state, one level up.  Its scan queues up behind whatever is in
        // flight on the handle's scan stream and behind the later call's detection (before_rerun: that call read the tail
        // columns this call's first scan already wrote -- the re-run writes the same values); other lanes' streams are left alone.
        const int from = c.mode_used;
        c.mode_used = level_up(h, from);
        // stay on that level for a while; every further failed probe doubles the while (a probe costs a wasted scan;
        // a call that climbs two levels counts once)
        h->auto_level = c.mode_used;
        if (!c.fell_back) {
            h->dense_sticky = h->sticky_len;
            h->sticky_len = std::min(h->sticky_len * 2, 1024);
        }
        c.fell_back = true;
        int rc = before_rerun(h, sl);
        if (rc == RT_OK) rc = enqueue_analysis(h, sl, c.mode_used, nullptr, from == RT_MODE_SPARSE && c.mode_used == RT_MODE_PREFILTER);
        if (rc != RT_OK) return rc;
        RT_HIP(h, hipEventSynchronize(sl.ev_done));
        flags = sl.h_counters[2];
    }
    // The call found more records than the slot's pinned pool holds (word 0 = records wanted): streams beyond its end
    // got truncated lists.  The pool grows and the call is analysed again on the level it ended on -- same buffer, same
    // look-back state, like a fall-back re-run -- so the caller loses nothing; later calls find the larger pool.  Not
    // possible for rt_extract (the caller's spectrogram is not kept) or when the host has no memory left: then the
    // truncated lists are delivered with RT_E_CAPACITY.
    // A stream wanted more records than the handle's per-stream capacity holds (word 4): the capacity grows and the call is
    // analysed again, like a call that outgrew the pool -- the reference has no limit (analyze.py:449-450).  Up to three times per
    // call (a re-run on a higher level may find more).  rt_extract cannot (the caller's spectrogram is not kept).
    // (Measured against the capacity the call's kernels RAN with: with two calls in flight the first one's growth may already cover what
    // the second one wanted -- its lists were still cut at the old capacity.)
    if ((flags & kFlagRecOverflow) && !c.is_extract && c.n_seg > 0 && sl.h_counters[4] > (unsigned long long)c.rec_cap_used && c.cap_grown < 3) {
        if (grow_record_capacity(h, sl.h_counters[4]) == RT_OK) {
            RT_TRACE_FETCH(h, sl, "capacity grown");
            ++c.cap_grown;
            for (int s = 0; s < h->cfg.n_streams; ++s) sl.h_overflow[s] = sl.h_incons[s] = 0;
            c.n_dense_streams = 0;
            int rc = before_rerun(h, sl);
            if (rc == RT_OK) rc = enqueue_analysis(h, sl, c.mode_used);
            if (rc != RT_OK) return rc;
            RT_HIP(h, hipEventSynchronize(sl.ev_done));
            flags = sl.h_counters[2];
            continue;
        }
    }
    if ((flags & kFlagRecOverflow) && !c.is_extract && c.n_seg > 0 && sl.h_counters[0] > (unsigned long long)sl.pool_cap &&
        sl.pool_cap < h->pool_max && c.pool_grown < 3) {
        if (grow_pool(h, sl, (int64_t)std::min<unsigned long long>(sl.h_counters[0], (unsigned long long)h->pool_max)) == RT_OK) {
            RT_TRACE_FETCH(h, sl, "pool grown");
            ++c.pool_grown;  // (up to three times: a partial dense re-run finds more records in its streams than the run the pool was sized from counted)
            for (int s = 0; s < h->cfg.n_streams; ++s) sl.h_overflow[s] = sl.h_incons[s] = 0;
            c.n_dense_streams = 0;
            int rc = before_rerun(h, sl);
            if (rc == RT_OK) rc = enqueue_analysis(h, sl, c.mode_used);
            if (rc != RT_OK) return rc;
            RT_HIP(h, hipEventSynchronize(sl.ev_done));
            flags = sl.h_counters[2];
            continue;
        }
    }
    break;
  }
    RT_TRACE_FETCH(h, sl, "settled");
    if (c.mode_used != RT_MODE_DENSE && !c.is_extract && c.n_seg > 0)
        for (int s = 0; s < h->cfg.n_streams; ++s) h->info.n_hot += sl.h_hot_total[s];
    // one wave per stream in the next calls' sparse detection while a stream holds half of what such a wave takes on average (heavier
    // batches -- BASELINE config 4: 1 500 cells per stream -- are faster with the per-list waves: most of their streams would be left to them anyway)
    if (c.mode_used != RT_MODE_DENSE && !c.is_extract && c.n_seg > 0) h->group_light = h->info.n_hot * 2 <= (int64_t)kGroupCells * h->cfg.n_streams;
    // (the exact pre-filter on input where it is not selective: see below)
    bool unselective = h->cfg.mode == RT_MODE_AUTO && c.mode_used == RT_MODE_RUNFILTER && !c.is_extract && c.n_seg > 0 && sl.h_seg_total &&
                       (int64_t)*sl.h_seg_total * 2 > (int64_t)h->cfg.n_streams * c.n_seg;
    // ... or a pre-filter level that went through with more than 1/32 of all cells on its candidate lists (possible where
    // hot_capacity was raised: signals whose side lobes fill every bin put 9 % of the cells there, and ordering lists of
    // 16 k cells per bucket took 14 - 18 ms per call where the dense path takes 1.9)
    if (h->cfg.mode == RT_MODE_AUTO && (c.mode_used == RT_MODE_RUNFILTER || c.mode_used == RT_MODE_PREFILTER) && !c.is_extract && c.n_seg > 0 &&
        h->info.n_hot * 32 > (int64_t)h->cfg.n_streams * c.n_seg * h->N)
        unselective = true;
    if (c.level_settled) {
        unselective = false;  // (settled on the first pass: the probe interval doubles once per call, not once per pass)
    } else if (h->cfg.mode == RT_MODE_AUTO && !c.is_extract && !c.fell_back && c.n_seg > 0 && level_rank(c.mode_used) < level_rank(h->auto_level) && !unselective) {
        // a probe of a lower level went through: the handle moves there (and from the pre-filter level it will
        // probe the plain sparse path after the usual interval)
        h->auto_level = c.mode_used;
        h->sticky_len = 16;
        h->dense_sticky = (c.mode_used == RT_MODE_SPARSE) ? 0 : 16;
    }
    if (unselective) {
        // The pre-filter went through, but more than half of all segments held cells it has to keep (or see above): its second scan
        // is then most of a scan, and the dense path (one scan, 16 B per sample) is faster -- measured at the reference's
        // default geometry with the noise floor 2 dB over the threshold: 213 k against 241 k MS/s.  The handle moves up like
        // after an overflow (without analysing this call again: its result stands) and probes this level later.
        h->auto_level = level_up(h, c.mode_used);
        h->dense_sticky = h->sticky_len;
        h->sticky_len = std::min(h->sticky_len * 2, 1024);
    }
    c.level_settled = true;
    if (flags & kFlagInconsistent) {
        std::memset(sl.h_incons, 0, (size_t)h->cfg.n_streams * sizeof(int32_t));
        h->err = "internal: candidate list lacks the cell preceding a run";
        if (peek) return kCallFailedInternal;
        c.pending = false;
        return RT_E_HIP;
    }
    if (h->timing && c.n_seg > 0) {
        (void)hipEventElapsedTime(&h->info.ms_stft, sl.ev_begin, sl.ev_scan);
        (void)hipEventElapsedTime(&h->info.ms_detect, sl.ev_scan, sl.ev_done);
        (void)hipEventElapsedTime(&h->info.ms_total, sl.ev_begin, sl.ev_done);
    }
    if (c.abs_counted && sl.h_abs_hot) {
        h->abs_hot_seen = *sl.h_abs_hot;
        h->abs_hot_valid = true;
    } else if (c.mode_used == RT_MODE_DENSE && c.n_dense_streams == 0) {
        h->abs_hot_valid = false;  // (the dense path keeps no count: what was seen is history by the time it is left)
    }
    h->info.mode_used = c.mode_used;
    h->info.fell_back = c.fell_back ? 1 : 0;
    h->info.n_dense_streams = c.n_dense_streams;

    const int S = h->cfg.n_streams;
    size_t total = 0;
    for (int s = 0; s < S; ++s) total += (size_t)sl.h_rec_count[s];
    h->info.n_records = (int64_t)total;
    *n_out = total;
    if (peek) {
        // nothing is delivered
    } else if (total && out && cap) {
        size_t w = 0;
        for (int s = 0; s < S && w < cap; ++s) {
            const int n = sl.h_rec_count[s];
            const int off = sl.h_rec_offset[s];
            for (int i = 0; i < n && w < cap; ++i) out[w++] = sl.h_records[(size_t)off + i];
        }
        c.pending = false;  // delivered
    } else if (total == 0) {
        c.pending = false;  // nothing to deliver
    }
    // (out == NULL / cap == 0 with records available is a size query: the call stays pending)
    if (flags & kFlagRecOverflow) {
        h->err = "record capacity exceeded (record_capacity per stream, or the record pool could not grow); results truncated";
        return RT_E_CAPACITY;
    }
    return RT_OK;
}

int rt_fetch(rt_handle *h, rt_record *out, size_t cap, size_t *n_out) {
    if (!h || !n_out) return RT_E_INVALID;
    if (h->kids.empty()) return fetch_one(h, out, cap, n_out, false);
    // lanes: size every lane first (a size query must leave all of them pending), then deliver in
    // stream order with the lane's first stream added to the records' stream index
    *n_out = 0;
    int truncated = RT_OK;
    size_t total = 0;
    int failed = RT_OK;
    for (rt_handle *k : h->kids) {
        size_t n = 0;
        const int rc = fetch_one(k, nullptr, 0, &n, true);
        if (rc != RT_OK && rc != RT_E_CAPACITY && failed == RT_OK) {
            h->err = k->err;
            failed = (rc == kCallFailed) ? RT_E_HOT_OVERFLOW : (rc == kCallFailedInternal) ? RT_E_HIP : rc;
        }
        total += n;
    }
    if (failed != RT_OK) {
        // one lane has no result for this call: the call is dropped in every lane, so that the lanes stay in step
        // (the next rt_fetch belongs to the next rt_process in all of them)
        for (rt_handle *k : h->kids) discard_oldest(k);
        *n_out = 0;
        return failed;
    }
    *n_out = total;
    h->info = rt_call_info{};
    const bool deliver = !(total && (!out || !cap));
    size_t w = 0;
    for (size_t i = 0; i < h->kids.size(); ++i) {
        rt_handle *k = h->kids[i];
        if (deliver && total && cap <= w) {
            // the caller's buffer is full: this lane's records are lost, but the call is consumed here as in the
            // lanes before it -- otherwise the lanes would be out of step from the next rt_fetch on
            discard_oldest(k);
        } else if (deliver) {
            size_t n = 0;
            const int rc = fetch_one(k, out ? out + w : nullptr, cap > w ? cap - w : 0, &n, false);
            if (rc == RT_E_CAPACITY) {
                truncated = rc;
                h->err = k->err;
            } else if (rc != RT_OK) {
                h->err = k->err;
                return rc;
            }
            const size_t got = (cap > w) ? (n < cap - w ? n : cap - w) : 0;
            if (out)
                for (size_t j = 0; j < got; ++j) out[w + j].stream += h->kid_base[i];
            w += got;
        }
        const rt_call_info &ki = k->info;
        h->info.n_seg = ki.n_seg;
        h->info.segs_per_chunk = ki.segs_per_chunk;
        h->info.mode_used = i == 0 ? ki.mode_used : (ki.mode_used < h->info.mode_used ? ki.mode_used : h->info.mode_used);
        h->info.fell_back |= ki.fell_back;
        h->info.n_dense_streams += ki.n_dense_streams;
        h->info.n_hot += ki.n_hot;
        h->info.n_records += ki.n_records;
        h->info.ms_stft += ki.ms_stft;      // sums over the lanes' launches (they overlap in time)
        h->info.ms_detect += ki.ms_detect;
        h->info.ms_total += ki.ms_total;
    }
    return truncated;
}

int rt_spectrogram(rt_handle *h, const void *iq_dev, int64_t n_samples, int64_t stream_stride, float *spec_dev) {
    if (!h || !iq_dev || !spec_dev) return RT_E_INVALID;
    if (!h->kids.empty())
        return for_each_lane(h, [&](rt_handle *k, int64_t s0) {
            const int64_t T = n_samples / k->N;
            return rt_spectrogram(k, static_cast<const char *>(iq_dev) + s0 * stream_stride * (int64_t)sizeof(cf), n_samples,
                                  stream_stride, spec_dev + s0 * T * k->N);
        });
    if (n_samples < 0 || n_samples > h->cfg.max_samples || stream_stride < n_samples ||
        reinterpret_cast<uintptr_t>(iq_dev) % 8u != 0 || reinterpret_cast<uintptr_t>(spec_dev) % 4u != 0) {
        h->err = "n_samples/stream_stride out of range for this handle, or misaligned pointer";
        return RT_E_INVALID;
    }
    RT_HIP(h, hipSetDevice(h->cfg.device));
    const int T = (int)(n_samples / h->N);
    if (T == 0) return RT_OK;
    RT_HIP(h, hipDeviceSynchronize());
    if (h->general) {
        launch_general(h, iq_dev, stream_stride, T, spec_dev, nullptr, false);
        RT_HIP(h, hipGetLastError());
        RT_HIP(h, hipStreamSynchronize(h->s_scan));
        return RT_OK;
    }
    StftParams sp = make_stft_params(h, h->slot[0], iq_dev, stream_stride, T, 0);
    sp.spec = spec_dev;
    launch_stft<2>(h, sp, h->cfg.n_streams * sp.blocks_per_stream, h->s_scan);
    RT_HIP(h, hipGetLastError());
    RT_HIP(h, hipStreamSynchronize(h->s_scan));
    return RT_OK;
}

int rt_calibrate_read(rt_handle *h, const void *iq_dev, int64_t n_samples, int64_t stream_stride) {
    if (!h || !iq_dev) return RT_E_INVALID;
    if (!h->kids.empty())
        return for_each_lane(h, [&](rt_handle *k, int64_t s0) {
            return rt_calibrate_read(k, static_cast<const char *>(iq_dev) + s0 * stream_stride * (int64_t)sizeof(cf), n_samples,
                                     stream_stride);
        });
    if (n_samples < 0 || n_samples > h->cfg.max_samples || stream_stride < n_samples || reinterpret_cast<uintptr_t>(iq_dev) % 8u != 0) {
        h->err = "n_samples/stream_stride out of range for this handle, or misaligned pointer";
        return RT_E_INVALID;
    }
    RT_HIP(h, hipSetDevice(h->cfg.device));
    const int T = (int)(n_samples / h->N);
    if (h->general || h->big) {
        h->err = "rt_calibrate_read: the load stream of the fused scans of nperseg 32 ... 4096 only";
        return RT_E_UNSUPPORTED;
    }
    if (T < 2) return RT_OK;
    RT_HIP(h, hipDeviceSynchronize());
    StftParams sp = make_stft_params(h, h->slot[0], iq_dev, stream_stride, T, 0);
    launch_stft<3>(h, sp, h->cfg.n_streams * sp.blocks_per_stream, h->s_scan);
    RT_HIP(h, hipGetLastError());
    RT_HIP(h, hipStreamSynchronize(h->s_scan));
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
