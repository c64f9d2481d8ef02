// rt_hostcheck.cpp -- HOST build of the scalar decision logic in rt_core.h, for
// unit tests only (tests/test_host_core.py).  It lets the CPU test-suite drive
// the exact functions the detect kernels execute (predicate, start walk,
// float64 duration gate, statistics, microsecond rounding, ordering + shadow
// verdict) against the oracle and the golden vectors without a GPU.
// It is NOT a fallback: the product library (rt_analyze.hip) never links or
// loads this file, and there is no FFT/STFT here at all.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/rt_analyze.h"
#include "rt_core.h"

using namespace rt;

extern "C" {

long long hc_timedelta_us(double seconds) { return (long long)timedelta_us(seconds); }

int hc_probe_stride(int nperseg, double fs, double min_d) { return probe_stride(nperseg, fs, min_d); }

double hc_seg_time(int k, int nperseg, double fs) { return seg_time(k, nperseg, fs); }

int hc_tail_cols(int nperseg, double fs, double max_d) {
    const double hop = seg_time(1, nperseg, fs) - seg_time(0, nperseg, fs);
    return (int)(max_d / hop) + 2;
}

// extract_signals + filter_shadow_signals for ONE stream on a dense,
// segment-major power map: spec[t*F + f].  last = previous map
// [n_seg_last][F] or NULL; tail_cols limits how far back `last` may be read
// (pass n_seg_last for the reference's unlimited look-back).
// Returns the number of records (written up to cap, ordered by (fi, start)).
int hc_extract(const float *spec, int n_seg, int n_bins, const float *last, int n_seg_last, int tail_cols,
               int nperseg, double fs, float thr, float snr, float cal_db, double min_d, double max_d,
               rt_record *out, int cap) {
    DetectParams p;
    p.n_seg = n_seg;
    p.n_seg_last = last ? n_seg_last : -1;
    p.tail_cols = last ? tail_cols : 0;
    p.stride = probe_stride(nperseg, fs, min_d);
    p.nperseg = nperseg;
    p.thr = thr;
    p.snr = snr;
    p.cal_db = cal_db;
    p.fs = fs;
    p.min_d = min_d;
    p.max_d = max_d;
    std::vector<rt_record> rec;
    std::vector<long long> ts, du;
    for (int fi = 0; fi < n_bins; ++fi) {
        auto cur = [&](int t) -> float { return spec[(size_t)t * n_bins + fi]; };
        auto prev = [&](int d) -> float { return last[(size_t)(n_seg_last - d) * n_bins + fi]; };
        float avg = 0.f;
        auto emit = [&](int start, int end, const RunStats &st) {
            rt_record r;
            std::memset(&r, 0, sizeof r);
            r.fi = fi;
            r.start = start;
            r.end = end;
            r.max_p = st.max_p;
            r.mean_p = st.mean_p;
            r.std_db = st.std_db;
            r.row_mean = avg;
            rec.push_back(r);
            ts.push_back(timedelta_us(start_time(p, start)));
            du.push_back(timedelta_us(run_duration(p, start, end)));
        };
        auto on_run = [&](int b, int e, float av) {
            avg = av;
            finish_run(p, b, e, av, cur, prev, emit);
        };
        scan_dense_row(p, cur, -1.0, &avg, on_run);
    }
    const int n = (int)rec.size();
    std::vector<rt_record> ordered(n);
    for (int i = 0; i < n; ++i) {
        int rank, shadow;
        rank_and_shadow(i, n, rec.data(), ts.data(), du.data(), cal_db, &rank, &shadow);
        ordered[rank] = rec[i];
        ordered[rank].shadowed = shadow;
    }
    for (int i = 0; i < n && i < cap; ++i) out[i] = ordered[i];
    return n;
}

// RT_MODE_AUTO's level bookkeeping (rt_core.h)
int hc_level_up(int prefilter_ok, int runfilter_ok, int mode) { return level_up(AutoLevels{prefilter_ok != 0, runfilter_ok != 0}, mode); }
int hc_level_down(int prefilter_ok, int runfilter_ok, int mode) { return level_down(AutoLevels{prefilter_ok != 0, runfilter_ok != 0}, mode); }
int hc_level_rank(int mode) { return level_rank(mode); }
int hc_probe_ruled_out(int target, int level, int valid, unsigned long long abs_hot, unsigned long long list_cells, unsigned long long cells_per_stream) {
    return probe_ruled_out(target, level, valid != 0, abs_hot, list_cells, cells_per_stream) ? 1 : 0;
}

// how often a cell of a chunk of L segments is counted by the sampled absolute-threshold bits (rt_core.h: abs_sampled):
// the sampled steps of i = 1 .. L under `phase`, times the period
int hc_abs_sample_period(int L) { return abs_sample_period(L); }
int hc_abs_sampled_weight(int L, int phase) {
    const int P = abs_sample_period(L);
    int n = 0;
    for (int i = 1; i <= L; ++i) n += abs_sampled(i, phase, P) ? P : 0;
    return n;
}
int hc_abs_sampled_segment(int L, int phase, int k) {  // the k-th sampled step (0-based) as a segment offset inside the chunk (seg - c0), or -1
    const int P = abs_sample_period(L);
    for (int i = 1; i <= L; ++i)
        if (abs_sampled(i, phase, P) && k-- == 0) return L - i;
    return -1;
}

int hc_minsum_group(int L, int gpw) { return minsum_group(L, gpw); }
double hc_minsum_margin(int m) { return (double)minsum_margin(m); }

// The planner of the exact run-length pre-filter (rt_core.h: RunPlanner, plan_tile_column -- the arithmetic of the plan_runs
// kernel) on one stream's threshold words hot[n_seg][w] (64-bit words): need[n_seg][w], tiled exactly as the kernel tiles a call
// of n_seg rows of lg = 4 w lanes (tile_rows = 0) or with the given tile length.  Returns the number of counter planes used.
int hc_plan_runs(const unsigned long long *hot, unsigned long long *need, int n_seg, int w, int r_in, int tile_rows) {
    const int r = r_in < n_seg + 1 ? r_in : n_seg + 1;  // (as the host side of the launch clamps it)
    const int B = tile_rows > 0 ? tile_rows : plan_tile_rows(n_seg, 4 * w, r);
    const int planes = r <= 16 ? 4 : r <= 256 ? 8 : 16;
    for (int a = 0; a < n_seg; a += B)
        for (int c = 0; c < w; ++c) {
            auto row = [&](int u) { return hot[(size_t)u * w + c]; };
            auto emit = [&](int t, unsigned long long v) { need[(size_t)t * w + c] = v; };
            if (planes == 4) plan_tile_column<4>(a, B, n_seg, r, row, emit);
            else if (planes == 8) plan_tile_column<8>(a, B, n_seg, r, row, emit);
            else plan_tile_column<16>(a, B, n_seg, r, row, emit);
        }
    return planes;
}
int hc_plan_tile_rows(int n_seg, int lg, int r) { return plan_tile_rows(n_seg, lg, r); }

}  // extern "C"
