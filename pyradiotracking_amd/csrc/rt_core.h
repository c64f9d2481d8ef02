// rt_core.h -- scalar decision logic of the analysis path, shared by the HIP
// kernels (device) and by the host-side unit-test harness (rt_hostcheck.cpp).
//
// Everything here is a restatement of reference arithmetic that must be
// decided bit-identically on the GPU:
//   * segment-centre times            scipy/signal/_spectral_py.py:2136-2137
//   * the "above" predicate            radiotracking/analyze.py:370, 378 (T10)
//   * start-of-plateau walk incl. look-back into the previous buffer
//                                      analyze.py:382-398 (T11-T13)
//   * duration gate in float64         analyze.py:420-433 (T14)
//   * plateau statistics               analyze.py:442-447 (T15)
//   * timedelta microsecond rounding   CPython Modules/_datetimemodule.c
//                                      (delta_new / accum), used by the shadow
//                                      filter analyze.py:300-311 (T16)
#ifndef RT_CORE_H
#define RT_CORE_H

#include <math.h>
#include <stdint.h>

#if defined(__HIP__)  // HIP translation units only (hipcc also compiles the plain C++ ones)
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

namespace rt {

// Geometry/thresholds of one detect pass (one spectrogram of T columns).
struct DetectParams {
    int32_t n_seg;        // T: columns of the current spectrogram
    int32_t n_seg_last;   // columns of the previous one, or -1 if there is none
    int32_t tail_cols;    // K: how many trailing columns of the previous one are readable
    int32_t stride;       // probe stride max(1, int(min_d / hop))         (analyze.py:354, 364)
    int32_t nperseg;      // N (for the time axis only)
    float thr;            // signal_threshold (linear)
    float snr;            // snr_threshold (linear)
    float cal_db;         // calibration (only to order maxima in the shadow filter)
    double fs;
    double min_d;         // seconds
    double max_d;         // seconds
};

// ---- RT_MODE_AUTO's levels (host-side bookkeeping of rt_analyze.hip; here so that the CPU suite can test it) ----
// Order: sparse < chunk-bit pre-filter (where the geometry allows it) < exact pre-filter (where its scratch exists) < dense.
// The mode numbers are rt_mode's (include/rt_analyze.h).
enum : int { kAutoDense = 1, kAutoSparse = 2, kAutoPrefilter = 3, kAutoRunfilter = 4 };
struct AutoLevels {
    bool prefilter_ok;  // the chunk-bit pre-filter exists at this geometry
    bool runfilter_ok;  // the exact pre-filter exists (and its scratch is allocated)
};
RT_HD int level_rank(int mode) { return mode == kAutoSparse ? 0 : mode == kAutoPrefilter ? 1 : mode == kAutoRunfilter ? 2 : 3; }
RT_HD int level_up(AutoLevels a, int mode) {
    if (mode == kAutoSparse && a.prefilter_ok) return kAutoPrefilter;
    if (level_rank(mode) < 2 && a.runfilter_ok) return kAutoRunfilter;
    return kAutoDense;
}
RT_HD int level_down(AutoLevels a, int mode) {
    if (mode == kAutoDense && a.runfilter_ok) return kAutoRunfilter;
    if (level_rank(mode) > 1 && a.prefilter_ok) return kAutoPrefilter;
    return kAutoSparse;
}
// Is a probe of `target` (= level_down of the handle's level) pointless?  `abs_hot` = the most cells at or above the
// absolute threshold any stream had in the last call a pre-filter level analysed (valid: there was one):
//   * the sparse lists hold `list_cells` (16 buckets x hot_capacity) cells per stream -- more than that cannot fit;
//   * the chunk-bit level needs chunks of L cells that do NOT all pass the absolute threshold: with a share q of the cells
//     over it a chunk bit is set with probability q^L in each of nperseg bins (q = 3/4, L = 32, 256 bins: 2.6 % of the
//     chunks), beyond that every chunk is kept.
RT_HD bool probe_ruled_out(int target, int level, bool valid, uint64_t abs_hot, uint64_t list_cells, uint64_t cells_per_stream) {
    if (!valid || target == level) return false;
    if (target == kAutoSparse) return abs_hot > list_cells;
    if (target == kAutoPrefilter && level == kAutoRunfilter) return cells_per_stream > 0 && abs_hot * 4u > 3u * cells_per_stream;
    return false;
}

// Chunks per "quiet level" sample of the exact pre-filter's per-bin thresholds (make_bin_thresholds): the quietest sum
// over `g` consecutive complete chunks of a workgroup's item, g the smallest power of two with g * L >= 32 segments (at
// most the item's gpw chunks) -- a small batch runs chunks of 4 segments, and the minimum over hundreds of 4-segment
// sums of exponentially distributed noise lies at a tenth of the mean, where the minimum over 32-segment sums lies at 0.55.
RT_HD int minsum_group(int L, int gpw) {
    int g = 1;
    while (g * L < 32 && g * 2 <= gpw) g *= 2;
    return g;
}

// Sampling of the absolute-threshold bits in the threshold-bit scan's items that need them only for AUTO's count (stft_scan, MODE 6
// with staged per-bin thresholds: abs_hot).  One step in P = abs_sample_period(L) builds them, P the largest power of two <= min(8, L)
// -- with chunks of 4 .. 7 segments (small batches) a fixed period of eight never sampled anything -- and a sampled cell counts P
// times.  Steps run i = 1 .. L in every chunk, so the phase must come from outside the chunk: `phase` = the item's number within
// its stream + the wave's number (uniform per wave), which turns from item to item whatever L is -- a pulse train whose period is
// a multiple of eight hops can neither hide from the count nor fill it.
RT_HD int abs_sample_period(int L) {
    int p = 1;
    while (p * 2 <= L && p < 8) p *= 2;
    return p;
}
RT_HD bool abs_sampled(int i, int phase, int period) { return ((i + phase) & (period - 1)) == 0; }

// Factor on the quiet-level estimate when its samples are longer than 32 segments (chunks of 37 .. 71 segments at nperseg >= 1024).
// The minimum over n samples of the mean of m exponentially distributed powers lies about 2 / sqrt(m) under the mean; the
// thresholds must stay under snr x the NEXT buffer's row mean, and with long samples (few of them, each close to the mean) that
// margin shrinks: 14 % at m = 128, where one bin in fifty failed the check in an experiment.  The factor puts every sample
// length on the footing of m = 32 (0.55 - 0.65 x the mean, measured safe).
RT_HD float minsum_margin(int m) {
    if (m <= 32) return 1.0f;
    return (1.0f - 2.0f / sqrtf(32.0f)) / (1.0f - 2.0f / sqrtf((float)m));
}

// times[k] of scipy: arange(N/2, B - N/2 + 1, N) / float(fs)
RT_HD double seg_time(int32_t k, int32_t nperseg, double fs) {
    return ((double)nperseg * 0.5 + (double)k * (double)nperseg) / fs;
}

// stride = max(1, int(min_d / (times[1] - times[0])))
RT_HD int32_t probe_stride(int32_t nperseg, double fs, double min_d) {
    double hop = seg_time(1, nperseg, fs) - seg_time(0, nperseg, fs);
    double q = min_d / hop;
    int32_t s = (q >= 2147483647.0) ? 2147483647 : (int32_t)q;  // int() truncates toward zero
    return s < 1 ? 1 : s;
}

// `not (p < thr) and not (p / avg < snr)` in float32
RT_HD bool cell_above(float p, float avg, float thr, float snr) {
    if (p < thr) return false;
    if (p / avg < snr) return false;
    return true;
}

// A maximal run [b, e) of above-cells is visited by the strided probe iff it
// contains a multiple of the stride (T9).  Returns that first probe or -1.
RT_HD int32_t first_probe_in_run(int32_t b, int32_t e, int32_t stride) {
    int32_t q = (b + stride - 1) / stride;
    int64_t ti = (int64_t)q * stride;
    return ti < e ? (int32_t)ti : -1;
}

// Outcome of the downward walk (analyze.py:382-398).
struct StartWalk {
    int32_t start;   // may be negative
    bool too_long;   // ran past the readable tail: duration certainly exceeds max_d
};

// `prev(d)` returns the power of the previous buffer's column n_seg_last-d
// (d >= 1, d <= tail_cols) for the bin at hand.
template <class PrevCell>
RT_HD StartWalk walk_start(const DetectParams &p, int32_t b, int32_t ti0, float avg, PrevCell prev) {
    StartWalk w;
    w.too_long = false;
    const int32_t start_min = (p.n_seg_last < 0) ? 0 : (1 - p.n_seg_last);
    if (ti0 <= start_min) {  // loop `while start > start_min` never runs
        w.start = ti0;
        return w;
    }
    // cells b..ti0 are above; b-1 (if >= 0) is not: the walk stops on it, or
    // earlier on start_min without testing (T11).
    int32_t s = b - 1;
    if (s < start_min) s = start_min;
    if (s >= 0 || b > 0) {
        w.start = s;
        return w;
    }
    // b == 0 and start_min < 0: continue into the previous buffer (T12)
    s = -1;
    for (;;) {
        if (s == start_min) break;  // not tested
        int32_t d = -s;
        if (d > p.tail_cols) {
            w.too_long = true;
            break;
        }
        if (!cell_above(prev(d), avg, p.thr, p.snr)) break;
        --s;
    }
    w.start = s;
    return w;
}

// start_dt / duration in float64 exactly as analyze.py:420-427
RT_HD double start_time(const DetectParams &p, int32_t start) {
    return start < 0 ? -seg_time(-start, p.nperseg, p.fs) : seg_time(start, p.nperseg, p.fs);
}
RT_HD double run_duration(const DetectParams &p, int32_t start, int32_t end) {
    return seg_time(end, p.nperseg, p.fs) - start_time(p, start);
}
RT_HD bool duration_ok(const DetectParams &p, double dur) {
    if (dur < p.min_d) return false;
    if (dur > p.max_d) return false;
    return true;
}

// np.max / np.mean / np.std(dB(.)) over the cells of a plateau.  `cell(i)`,
// i in [0, n), yields the i-th element of `data` (analyze.py:437-440).
struct RunStats {
    float max_p, mean_p, std_db;
};

RT_HD float db10(float v) { return 10.0f * log10f(v); }

// Canonical summation order (so the wave-cooperative device code, the dense
// kernel and the host check agree bit for bit): 64 interleaved partial sums
// (cell k goes to partial k mod 64, in k order), folded by halving
// (p[l] += p[l + off], off = 32, 16, ... 1).  Sums run in float64 over the
// float32 values np.mean / np.std see; np.max propagates NaN.
constexpr int kStatLanes = 64;

template <class Cell>
RT_HD RunStats run_stats(int32_t n, Cell cell) {
    double ps[kStatLanes], pd[kStatLanes];
    float pm[kStatLanes];
    bool any_nan = false;
    for (int l = 0; l < kStatLanes; ++l) {
        ps[l] = 0.0;
        pd[l] = 0.0;
        pm[l] = -INFINITY;
    }
    for (int32_t k = 0; k < n; ++k) {
        const int l = k & (kStatLanes - 1);
        const float v = cell(k);
        ps[l] += (double)v;
        pd[l] += (double)db10(v);
        if (v != v) any_nan = true;
        if (v > pm[l]) pm[l] = v;
    }
    for (int off = kStatLanes / 2; off > 0; off >>= 1)
        for (int l = 0; l < off; ++l) {
            ps[l] += ps[l + off];
            pd[l] += pd[l + off];
            if (pm[l + off] > pm[l]) pm[l] = pm[l + off];
        }
    const double mean_db = pd[0] / (double)n;
    double pa[kStatLanes];
    for (int l = 0; l < kStatLanes; ++l) pa[l] = 0.0;
    for (int32_t k = 0; k < n; ++k) {
        const double d = (double)db10(cell(k)) - mean_db;
        pa[k & (kStatLanes - 1)] += d * d;
    }
    for (int off = kStatLanes / 2; off > 0; off >>= 1)
        for (int l = 0; l < off; ++l) pa[l] += pa[l + off];
    RunStats r;
    r.max_p = any_nan ? NAN : pm[0];
    r.mean_p = (float)(ps[0] / (double)n);
    r.std_db = (float)sqrt(pa[0] / (double)n);
    return r;
}

// datetime.timedelta(seconds=x) -> whole microseconds, CPython's algorithm:
// split off the integer seconds exactly, scale the fraction by 1e6 in double,
// split again, round the leftover half-to-even against the parity of the sum.
RT_HD int64_t timedelta_us(double seconds) {
    double ip;
    double frac = modf(seconds, &ip);
    int64_t us = (int64_t)ip * 1000000LL;
    if (frac == 0.0) return us;
    double ip2;
    double left = modf(1000000.0 * frac, &ip2);
    us += (int64_t)ip2;
    if (left != 0.0) {
        double whole = round(left);
        if (fabs(whole - left) == 0.5) {
            int odd = (int)(us & 1LL);
            whole = 2.0 * round((left + odd) * 0.5) - odd;
        }
        us += (int64_t)whole;
    }
    return us;
}

// A maximal run [b, e) of above-cells of the current buffer -> at most one
// plateau (analyze.py:401-433 in run-based form): the
// cheap decisions.  `prev(d)` reads the previous buffer's cell n_seg_last - d.
// Returns true and the first cell of `data` if the run becomes a signal.
template <class Prev>
RT_HD bool gate_run(const DetectParams &p, int32_t b, int32_t e, float avg, Prev prev, int32_t *start_out) {
    if (e == p.n_seg) return false;  // laps into the next buffer (analyze.py:415)
    const int32_t ti0 = first_probe_in_run(b, e, p.stride);
    if (ti0 < 0) return false;       // no strided probe lands in the run (T9)
    const StartWalk sw = walk_start(p, b, ti0, avg, prev);
    if (sw.too_long) return false;
    if (!duration_ok(p, run_duration(p, sw.start, e))) return false;
    *start_out = sw.start;
    return true;
}

// gate + statistics for one run, sequentially (host check; the kernels gate
// per thread and compute the statistics wave-cooperatively in the same order)
template <class Cur, class Prev, class Emit>
RT_HD void finish_run(const DetectParams &p, int32_t b, int32_t e, float avg, Cur cur, Prev prev, Emit emit) {
    int32_t start;
    if (!gate_run(p, b, e, avg, prev, &start)) return;
    auto cell = [&](int32_t k) -> float {
        const int32_t t = start + k;
        return t < 0 ? prev(-t) : cur(t);
    };
    emit(start, e, run_stats(e - start, cell));
}

// Sequential scan of one bin's row of a dense spectrogram (analyze.py:357-450):
// calls on_run(b, e, avg) for every maximal run of above-cells.  Returns false
// when no cell reaches the absolute threshold (the row mean is then unused).
// `row_sum` < 0 means "not known": the row is summed here.
template <class Cur, class OnRun>
RT_HD bool scan_dense_row(const DetectParams &p, Cur cur, double row_sum, float *avg_out, OnRun on_run) {
    const int32_t T = p.n_seg;
    double sum = 0.0;
    bool any = false;
    for (int32_t t = 0; t < T; ++t) {
        const float v = cur(t);
        sum += (double)v;
        any |= !(v < p.thr);
    }
    if (!any) return false;
    if (row_sum >= 0.0) sum = row_sum;
    const float avg = (float)sum / (float)T;  // np.mean(row) (analyze.py:375)
    *avg_out = avg;
    int32_t b = -1;
    for (int32_t t = 0; t <= T; ++t) {
        const bool ab = (t < T) && cell_above(cur(t), avg, p.thr, p.snr);
        if (ab) {
            if (b < 0) b = t;
            continue;
        }
        if (b < 0) continue;
        const int32_t rb = b;
        b = -1;
        on_run(rb, t, avg);
    }
    return true;
}

// is_shadow_of (analyze.py:300-311) on microsecond offsets from ts_start.
RT_HD bool shadowed_by(int64_t ts_i, int64_t dur_i, float max_i, int64_t ts_j, int64_t dur_j, float max_j) {
    if (ts_i > ts_j + dur_j) return false;
    if (ts_i + dur_i < ts_j) return false;
    return max_j > max_i;
}

// Position of record i in (fi, start) order and its shadow verdict against
// the unfiltered list (analyze.py:325).  max is compared as the reference's
// float32 dBW figure (analyze.py:442).
template <class Rec>
RT_HD void rank_and_shadow(int32_t i, int32_t n, const Rec *rec, const long long *ts_us, const long long *dur_us,
                           float cal_db, int32_t *rank_out, int32_t *shadow_out) {
    const int32_t fi = rec[i].fi, st = rec[i].start;
    const float mx_i = db10(rec[i].max_p) - cal_db;
    int32_t rank = 0, shadow = 0;
    for (int32_t j = 0; j < n; ++j) {
        if (rec[j].fi < fi || (rec[j].fi == fi && rec[j].start < st)) ++rank;
        const float mx_j = db10(rec[j].max_p) - cal_db;
        if (shadowed_by(ts_us[i], dur_us[i], mx_i, ts_us[j], dur_us[j], mx_j)) shadow = 1;
    }
    *rank_out = rank;
    *shadow_out = shadow;
}

// ---- exact run-length pre-filter: the planner's arithmetic (rt_kernels.h: plan_runs; host copy in rt_hostcheck.cpp) ----
// One 64-bit word = 64 independent cells (the threshold bits of four lanes' sixteen bins each); the planner walks a word column
// upwards in time with two bit-sliced counters of K planes, r = the cells a plateau needs:
//     SAT[u] = the threshold run ending at row u has r cells or more     (sticky while the run lasts)
//     FAR[u] = no SAT row among the last r rows
// step(h) takes row u's word and returns C[u - r + 1] = ~FAR[u]: "row u - r + 1 lies in a threshold run of >= r cells".
// K planes hold counts up to 2^K - 1 >= r - 1; a count that wraps while its flag is already set changes nothing.
constexpr int kPlanRowsPerWave = 16384;  // rows of one wave at most (its byte flags in LDS)
constexpr int kPlanMaxRun = 65536;       // r at most (16 counter planes)
template <int K>
struct RunPlanner {
    unsigned long long c1[K], c2[K], inv[K], sat, far;
    RT_HD void init(int r) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            c1[k] = 0ull;
            c2[k] = 0ull;
            inv[k] = (((unsigned)(r - 1) >> k) & 1u) ? 0ull : ~0ull;  // (plane ^ inv[k]) is all ones where the plane agrees with bit k of r - 1
        }
        sat = 0ull;
        far = ~0ull;
    }
    RT_HD unsigned long long step(unsigned long long hh) {
        unsigned long long eq = ~0ull;
#pragma unroll
        for (int k = 0; k < K; ++k) eq &= c1[k] ^ inv[k];
        sat = hh & (sat | eq);  // run[u - 1] reached r - 1 at some point and the bit stayed set
        unsigned long long carry = ~0ull;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const unsigned long long t = c1[k] ^ carry;
            carry &= c1[k];
            c1[k] = t & hh;  // (a clear bit ends the run)
        }
        eq = ~0ull;
#pragma unroll
        for (int k = 0; k < K; ++k) eq &= c2[k] ^ inv[k];
        far = ~sat & (far | eq);
        carry = ~0ull;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const unsigned long long t = c2[k] ^ carry;
            carry &= c2[k];
            c2[k] = t & ~sat;
        }
        return ~far;
    }
};
// Rows per planning tile for a call of n_seg segments: tiles a few halos long (a tile reads 2 r - 1 rows beside its own),
// a whole number of waves' worth of them per stream (a wave holds 64 / w tiles side by side), a wave's rows within its flags.
RT_HD int plan_tile_rows(int n_seg, int lg, int r) {
    const int tpw = 64 / (lg / 4);
    const int target = (4 * r > 64) ? 4 * r : 64;
    int waves = n_seg / (tpw * target);
    if (waves < 1) waves = 1;
    const int waves_min = (n_seg + kPlanRowsPerWave - 1) / kPlanRowsPerWave;
    if (waves < waves_min) waves = waves_min;
    const int tiles = waves * tpw;
    int rows = (n_seg + tiles - 1) / tiles;
    return rows < 1 ? 1 : rows;
}
// One word column of one tile (rows a .. a + B - 1 of a buffer of n_seg rows): `row(u)` yields row u's word for 0 <= u < n_seg,
// `emit(t, need)` receives need[t] = C[t] | C[t + 1] (the cell before a run: `data` starts on it) for the tile's rows inside
// the buffer.  Rows before the buffer count as set (a run through t = 0 may continue a plateau of the previous buffer: any
// length keeps it), rows past it as clear.  The kernel runs the same steps, eight rows per batch with the loads ahead.
template <int K, class Row, class Emit>
RT_HD void plan_tile_column(int a, int B, int n_seg, int r, Row row, Emit emit) {
    RunPlanner<K> pl;
    pl.init(r);
    unsigned long long c_prev = 0ull;
    const int t_end = (a + B < n_seg) ? a + B : n_seg;
    for (int u = a - r + 1; u <= a + B + r - 1; ++u) {
        const unsigned long long hh = (u < 0) ? ~0ull : (u < n_seg) ? row(u) : 0ull;
        const unsigned long long c_now = pl.step(hh);
        const int t = u - r;
        if (t >= a && t < t_end) emit(t, c_prev | ((t + 1 < n_seg) ? c_now : 0ull));
        c_prev = c_now;
    }
}

}  // namespace rt
#endif
