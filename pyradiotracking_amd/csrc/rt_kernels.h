// rt_kernels.h -- gfx950 kernels of the analysis path.
//
//   stft_scan<R3, MODE>   IQ -> windowed, detrended segment FFT -> power, fused
//                         with per-bin partial row sums, the dense look-back
//                         tail and either sparse candidate emission (MODE 0),
//                         a dense spectrogram (MODE 1) or the spectrogram only
//                         (MODE 2, debug; MODE 3 = loads only, for PMC traffic
//                         calibration; MODE 4 / 5 = the two passes of the
//                         run-length pre-filter, see below).  Replaces scipy.signal.spectrogram
//                         as called at radiotracking/analyze.py:234-241.
//   detect_sparse         per stream: finish row means, sort candidates,
//                         plateau extraction + statistics + shadow filter.
//   detect_dense          the same on a dense spectrogram.
//                         Both replace analyze.py:330-452 and :282-328.
//
// Segment FFT: N = 256*R3 points as radix passes 16 x 16 x R3 over a "lane
// group" of LG = N/16 lanes holding 16 points each.  With n = a + LG*m and
// k = k1 + 16*q1 + 256*q2:
//   pass 1 (in-lane over m)          A[a][k1]  = sum_m x[a+LG*m] W16^(m k1),  times W_N^(a k1)
//   exchange 1 (LDS, [k1][b][c], a = b + R3*c)
//   pass 2 (in-lane over c)          B[b][k1][q1] = sum_c A[b+R3*c][k1] W16^(c q1), times W_LG^(b q1)
//   exchange 2 (LDS, [k1][q1][b])    (R3 > 1 only)
//   pass 3 (in-lane over b)          X[k1+16*q1+256*q2] = sum_b B[b][k1][q1] W_R3^(b q2)
// For N = 256 a lane group is 16 lanes (four segments per wave64) and the whole
// transform needs one LDS exchange; no barrier is needed while LG <= 64 because
// a group then lives inside one wave.
#ifndef RT_KERNELS_H
#define RT_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/rt_analyze.h"
#include "rt_diag.h"  // first: refuses laboratory switches in a product build
#include "rt_core.h"
#include "rt_fft.h"

namespace rt {

constexpr int kBlock = 256;          // threads per workgroup (4 waves)
constexpr int kRowF2 = 18;           // LDS exchange row stride in float2 (16 data + 2 pad = 144 B)

struct StftParams {
    const void *iq;          // [S][stream_stride] complex64, or interleaved uint8 I/Q (U8 instantiations)
    int64_t stream_stride;   // samples
    int32_t n_streams;
    int32_t n_seg;           // T
    int32_t segs_per_chunk;  // L
    int32_t chunks;          // ceil(T / L) chunks per stream
    int32_t blocks_per_stream;
    int32_t tail_cols;       // K
    const float *window;     // [N] window coefficients times sqrt(scale)
    const float *window_t;   // nperseg 4096: the same in the order the lanes read it.  stft_scan64: [n0][jq][lane][e] = window[lane + 64 m],
                             // m = n0 + 4 (4 jq + e) (16-byte pieces of the quarter n0 of a lane's 64 elements); stft_scan<16>: [lane][m] = window[lane + 256 m]
    const cf *tw1;           // [LG][16]   W_N^(a*k1); stft_scan64: [16][64] rows 0..6 W_N^(8 l d), d = 1..7, rows 7..13 W_N^(l c), c = 1..7
    const cf *tw2;           // [R3][16]   W_LG^(b*q1)
    float scale;
    float thr;
    float lin_c[3];          // LIN instantiations: W[0]/N, W[1]/N, W[N-1]/N of the (scaled) window's transform (real)
    const float *thr_s;      // [S] per-stream thresholds (one calibration per SDR, analyze.py:115), or null: `thr` for all
    float *psum;             // [S][blocks_per_stream][N] partial row sums (one row per workgroup)
    float *tail;             // [S][K][N] trailing K columns (written)
    float *spec;             // MODE 1/2: [S][T][N]
    uint2 *hot;              // MODE 0: [S][kBuckets][hot_cap] (key = bin << tbits | t, bits of P), bucket = bin & (kBuckets-1)
    uint32_t *hot_count;     // MODE 0: [S][kBuckets]
    int32_t hot_cap;         // cells per (stream, bucket)
    int32_t tbits;           // bits reserved for t in a key (2^tbits >= T)
    const int32_t *stream_list;  // null, or the n_streams streams this launch analyses (AUTO's dense re-run of the few streams
                             // whose candidate lists overflowed): the grid's stream index is a position in this list, per-stream
                             // arrays are indexed by the stream it names, the dense spectrogram by the position
    uint16_t *full;          // [S][chunks][LG] per lane: bit r = "every cell of this chunk in the lane's bin r passes the
                             // absolute threshold".  Written by MODE 0 / 4, read by MODE 5.
    uint16_t *first;         // [S][segs_per_chunk][LG] per lane: MODE 0 / 4 write bit r = "the cell (t, lane's bin r) passes it" for
                             // the segments t of chunk 0; plan_pass_b turns that into "every cell 0 .. t does"; read by MODE 5
    int32_t *item_chunks;    // [S][blocks_per_stream][lane groups per workgroup] MODE 5: the chunks a workgroup transforms (plan_pass_b;
                             // `chunks` = none), item_count[s] workgroups per stream
    int32_t *item_count;     // [S]
    uint32_t *work;          // two words, zero between launches: tickets drawn for further items, workgroups that have left
    // exact run-length pre-filter (MODE 6 / plan_runs / MODE 7, see plan_runs)
    uint16_t *cell_hot;      // [S][T][LG] per lane: bit r = "cell (t, lane's bin r) passes the absolute threshold" (MODE 6 writes)
    const uint16_t *cell_need;  // [S][T][LG] the cells the selective pass emits (plan_runs writes, MODE 7 reads)
    const int32_t *seg_list; // [S][T] the segments that hold such cells, in any order; seg_count[s] of them
    const int32_t *seg_count;
    uint32_t *abs_hot;       // [S] (+ word [S]: their maximum, by the planning kernel), or null: MODE 4 / 6 add the stream's cells at or above the
                             // absolute threshold -- what the sparse lists would have to hold at least (AUTO skips probes that cannot succeed)
    uint32_t *chunk_min;     // [S][blocks_per_stream][N] float bits (one row per work item, beside psum's), or null: per bin the smallest sum of P over a
                             // complete chunk (group of chunks: minsum_group) of the item, 0x7f7f7f7f where it holds none -- the quiet level of the bin, for
                             // the next call's thr_bin (make_bin_thresholds takes the smallest over a stream's items; plain stores: one atomicMin per bin
                             // and item on a [S][N] array cost the threshold-bit scan 5 %, profiles/r05_c_*)
    const float *thr_bin;    // MODE 6, or null: [S][LG][16] a second, per-bin threshold in lane order; a cell's bit is set only if it passes both
    // LIN instantiations, guard of the detrend by linearity: a stream whose constant offset lies more than 60 dB over its quietest
    // bin's per-sample power is marked (host-visible word); the host analyses the call again with that stream -- and only that
    // stream, from then on -- on the subtract-first kernels (rt_fetch).  With D = the sum over an item's segments of
    // |sum of the samples|^2 (N^2 x the power the mean subtraction removes), an item marks its stream when
    // D > dc_limit x (smallest of its row sums, the bins 0 and +-1 aside)  [dc_limit = 1e6 N^2 fs: the offset 60 dB over the noise]
    // and  D > dc_limit2 x (sum of its row sums)  [dc_limit2 = 100 N fs: the offset holds 20 dB more power than everything else in
    // those segments -- a tag a fraction of a bin from the centre frequency also has a large segment mean, but what the mean
    // subtraction leaves of it is as strong as what it takes, and the round-off of a strong tone is in every bin in either form].
    int32_t *dc_flag;        // [S], or null: no guard
    float dc_limit, dc_limit2;
    const int32_t *sub_first;  // LIN instantiations, or null: [S] non-zero = a marked stream: this launch leaves it alone (its items end at
                             // once), a second launch of the subtract-first instantiation over `stream_list` = the marked streams takes it
    int32_t spec_by_stream;  // MODE 1 with a stream list: the dense spectrogram is indexed by stream, not by position in the list
    // stft_scan64 (its items are drawn per wave, latest chunks first): the EARLIEST short_chunks chunks of a stream -- the items
    // drawn last -- have short_len segments instead of segs_per_chunk, so that the launch's last items are short (chunk_span).
    // 0: every chunk has segs_per_chunk segments.
    int32_t short_chunks, short_len;
#ifdef RT_STAMPS
    uint32_t *dbg;           // [workgroups][4 waves][kStamps] cycles per stage, [kStamps - 1] = steps taken
#endif
};

// Run-length pre-filter (inputs whose noise crosses the absolute threshold, so that MODE 0 overflows its candidate
// lists).  A run can only become a signal if it is at least `stride - 1` cells long (rt_core.h: gate_run, the duration
// gate) or reaches back into the previous buffer (then it contains t = 0).  With chunks of L <= stride / 2 segments a
// run of that length covers at least one aligned chunk completely.  So:
//   pass A (MODE 4, or MODE 0 itself before it overflowed): per (stream, chunk, bin) one bit "all L cells >= thr",
//          and the threshold bits of chunk 0 cell by cell (a run through t = 0 either ends inside chunk 0 or makes
//          chunk 0 all hot; prefix_first and-s them up from t = 0, which leaves exactly the cells of such runs),
//          next to everything a scan writes (row sums, look-back tail);
//   pass B (MODE 5): only the chunks with a set bit in themselves or a neighbour are transformed again, and only the
//          flagged bins emit candidate cells.  Every chunk a qualifying run touches is flagged or next to a flagged
//          one, so the detect kernels see those runs complete; what they see of other runs is too short to pass the
//          duration gate (DetectArgs::filtered: a run without its preceding cell is dropped, not an error).
// In noise with P(cell >= thr) = p a chunk bit is set with probability p^L: at L = 32 pass B touches nothing but the
// neighbourhoods of real signals until the threshold sits ~7 dB under the noise floor.

constexpr int kBuckets = 16;  // candidate lists per stream: bucket = bin & 15 (a bin never spans buckets).
                              // 64 was measured: detect -6 %, but the scan +2 % (N=256) .. +8 % (N=1024)

constexpr int kStageCap = 128;  // candidate cells staged per wave before a flush (1 KiB)

// P[r] for a register index that differs from lane to lane: a binary tree of selects on the bits of r (CNT - 1 v_cndmask; written
// as a recursion on scalars -- with local arrays for the levels hipcc turned the tree into an indexed load from scratch)
template <int LO, int CNT, int NP>
__device__ __forceinline__ float pick_range(const float (&P)[NP], int r) {
    if constexpr (CNT == 1) {
        return P[LO];
    } else {
        const float lo = pick_range<LO, CNT / 2>(P, r), hi = pick_range<LO + CNT / 2, CNT / 2>(P, r);
        return (r & (CNT / 2)) ? hi : lo;
    }
}

// bits * 16 + the four bits !(p3 < t3), !(p2 < t2), !(p1 < t1), !(p0 < t0) (p3's the highest; a NaN power passes, as for the
// reference's `not (P < thr)`): a compare into a scalar pair and an add-with-carry `bits + bits + carry` per cell -- two vector
// instructions where compare -> select 0 / 1 -> shift / or takes three.  gfx950 wants two other instructions between a compare
// and the first reader of its scalar result (hipcc puts `s_nop 1` there): three scalar pairs in rotation provide them.
__device__ __forceinline__ uint32_t shift_in4(uint32_t bits, float p0, float p1, float p2, float p3, float t0, float t1, float t2, float t3) {
    unsigned long long m0, m1, m2;
    asm("v_cmp_nlt_f32_e64 %1, %7, %11\n\t"
        "v_cmp_nlt_f32_e64 %2, %6, %10\n\t"
        "v_cmp_nlt_f32_e64 %3, %5, %9\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, %0, %1\n\t"
        "v_cmp_nlt_f32_e64 %1, %4, %8\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, %0, %2\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, %0, %3\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, %0, %1"
        : "+v"(bits), "=&s"(m0), "=&s"(m1), "=&s"(m2)
        : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(t0), "v"(t1), "v"(t2), "v"(t3)
        : "vcc");
    return bits;
}

// Append a wave's staged candidate cells to the 16 per-bucket lists of its
// stream (bucket = bin & (kBuckets-1)).  Two passes over the <= kStageCap staged cells:
// count per bucket (ballots), lanes 0..15 reserve their bucket's slots with one
// returned atomic each, then every cell is stored at base + rank.
__device__ __forceinline__ void flush_stage(const StftParams &p, int s, const uint2 *stg, int n) {
    const int lane = threadIdx.x & 63;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // rare path: loops are deliberately not unrolled (register pressure of the hot loop matters more)
    uint32_t my_cnt = 0;  // lane b (< 16): cells of bucket b
#pragma nounroll
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const int bk = (i < n) ? (int)((stg[i].x >> p.tbits) & (kBuckets - 1)) : kBuckets;
#pragma nounroll
        for (int b = 0; b < kBuckets; ++b) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(bk == b);
            if (lane == b) my_cnt += (uint32_t)__builtin_popcountll(m);
        }
    }
    uint32_t slot_base = 0;  // lane b: first free slot of bucket b for this flush
    if (lane < kBuckets && my_cnt) slot_base = atomicAdd(&p.hot_count[s * kBuckets + lane], my_cnt);
#pragma nounroll
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const uint2 e = (i < n) ? stg[i] : make_uint2(0u, 0u);
        const int bk = (i < n) ? (int)((e.x >> p.tbits) & (kBuckets - 1)) : kBuckets;
#pragma nounroll
        for (int b = 0; b < kBuckets; ++b) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(bk == b);
            if (m == 0) continue;
            const uint32_t first = __shfl(slot_base, b, 64);
            if (bk == b) {
                const uint32_t slot = first + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
                if (slot < (uint32_t)p.hot_cap) p.hot[((int64_t)s * kBuckets + b) * p.hot_cap + slot] = e;
            }
            if (lane == b) slot_base += (uint32_t)__builtin_popcountll(m);
        }
    }
    __builtin_amdgcn_wave_barrier();
}

template <int LG>
__device__ __forceinline__ void group_sync() {
#ifdef RT_EXP_NOBAR1  // timing-only diagnostic: the exchange without its workgroup barrier (wrong spectra)
    if constexpr (false) {
#else
    if constexpr (LG > 64) {
#endif
        __syncthreads();
    } else {
        // a lane group lives inside one wave: DS operations of a wave execute
        // in program order, only the compiler must not reorder across this.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// ordering point for LDS traffic that stays inside one wave (DS operations of a wave execute in program
// order; only the compiler must not move them across)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// v + (v of another lane of the same row of 16), the lane picked by a DPP control: the shuffle
// rides on the add's operand fetch, no LDS round trip (ds_bpermute costs ~100 cycles of latency
// per step, four dependent steps per segment)
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true);
    return v + __int_as_float(moved);
}

// v + (lane 15 of the previous row | lane 31) for the rows in ROW_MASK (DPP row_bcast15 / row_bcast31), v elsewhere
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add_rows(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
    return v + __int_as_float(moved);
}

// Sum over the lanes of a lane group that sit in this wave, the same bits in every lane of the group (LG <= 64: the
// group's sum; LG > 64: this wave's share of it).  Within a row of 16 lanes: quad butterflies (quad_perm [1,0,3,2],
// [2,3,0,1]), then row_half_mirror and row_mirror fold the other quad / other half in.  Across rows: row_bcast15 adds
// a row's sum into the next row, row_bcast31 the lower half's into the upper rows, and the row that ends up with the
// total hands it out through a scalar register (v_readlane) -- no LDS round trip: the two dependent ds_bpermute steps
// this replaces cost the wave ~250 cycles per step at nperseg 1024 and, with the workgroup exchange behind them,
// ~1 000 at nperseg 4096 (profiles/r03_d_stage_stamps.txt, stage 3).
template <int LG>
__device__ __forceinline__ cf wave_sum(cf v) {
    // (lane groups of 2 / 4 / 8 lanes -- nperseg 32 / 64 / 128 -- stop after the first one / two / three folds)
    v.x = dpp_add<0xB1>(v.x);   v.y = dpp_add<0xB1>(v.y);    // quad_perm [1,0,3,2]
    if constexpr (LG >= 4) { v.x = dpp_add<0x4E>(v.x);   v.y = dpp_add<0x4E>(v.y); }    // quad_perm [2,3,0,1]
    if constexpr (LG >= 8) { v.x = dpp_add<0x141>(v.x);  v.y = dpp_add<0x141>(v.y); }   // row_half_mirror
    if constexpr (LG >= 16) { v.x = dpp_add<0x140>(v.x);  v.y = dpp_add<0x140>(v.y); }  // row_mirror
    if constexpr (LG >= 32) {
        v.x = dpp_add_rows<0x142, 0xA>(v.x);  v.y = dpp_add_rows<0x142, 0xA>(v.y);  // rows 1, 3 += rows 0, 2
        if constexpr (LG >= 64) {
            v.x = dpp_add_rows<0x143, 0xC>(v.x);  v.y = dpp_add_rows<0x143, 0xC>(v.y);  // rows 2, 3 += (rows 0 + 1)
            v.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.x), 63));
            v.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.y), 63));
        } else {
            // two groups of 32 lanes: their sums sit in rows 1 and 3
            const float ax = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.x), 31));
            const float ay = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.y), 31));
            const float bx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.x), 63));
            const float by = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.y), 63));
            const bool upper = (threadIdx.x & 32) != 0;
            v.x = upper ? bx : ax;
            v.y = upper ? by : ay;
        }
    }
    return v;
}

// all-reduce (sum) over the LG lanes of a lane group
template <int LG>
__device__ __forceinline__ cf group_sum(cf v, cf *red /* [kBlock/64] LDS */) {
    v = wave_sum<LG>(v);
    if constexpr (LG > 64) {
        // One workgroup barrier.  `red` needs no barrier before the write: its readers of the previous step
        // all read it before they reached that step's exchange barrier, which this wave has passed since.
        // The barrier also separates the previous step's last use of the exchange rows from this step's.
        const int wave = threadIdx.x >> 6;
        constexpr int WPG = LG / 64;  // waves per group
        if ((threadIdx.x & 63) == 0) red[wave] = v;
#ifndef RT_EXP_NOBAR0  // timing-only diagnostic: without the barrier (wrong sums, exchange rows unprotected)
        __syncthreads();
#endif
        const int w0 = (wave / WPG) * WPG;
        cf s{0.f, 0.f};
#pragma unroll
        for (int i = 0; i < WPG; ++i) s = cadd(s, red[w0 + i]);
        v = s;
    }
    return v;
}

// Exchange layouts for R3 > 1 (tools/lds_banks.py models them with the gfx950 bank rules: every ds_write_b64 and
// ds_read_b128 of both exchanges is conflict-free; the plain [row][column] layout was 2-way conflicted on every store).
//   exchange 1: element (a = b + R3*c, k1) goes to row k1*R3 + b, physical column (c + s1(b)) & 15 with
//               s1(b) = (16/R3 - 2) * b.  The reader takes the row as it lies: a sequence rotated by s is a phase
//               W16^(s*q1) on its transform, and that phase is folded into the pass-2 twiddle table (host side).
//   exchange 2: the R3-wide column groups u of the rows of one k1 are rotated by sh(k1) = (k1*R3/8) mod (16/R3);
//               pass 3 transforms inside the groups, so the rotation only renumbers its output registers (bin_of).
template <int R3>
__device__ __forceinline__ int x1_rotation(int b) {
    return ((16 / R3 - 2) * b) & 15;
}
template <int R3>
__device__ __forceinline__ int x2_rotation(int k1) {
    return (k1 * R3 / 8) & (16 / R3 - 1);
}

// bin index of result register r in lane `lt` of a group after the last pass
// (LGv: lanes of a group -- 16 R3, or 8 / 4 / 2 at nperseg 128 / 64 / 32, where R3 = 1 and the last pass is over the group's lanes alike)
template <int R3, int LGv = 16 * R3>
__device__ __forceinline__ int bin_of(int lt, int r) {
    if constexpr (R3 == 1) {
        return lt + LGv * r;  // k1 = lt, q1 = r
    } else {
        constexpr int G = 16 / R3;
        const int k1 = lt / R3, qg = lt % R3;
        const int u = (r / R3 - x2_rotation<R3>(k1)) & (G - 1), q2 = r % R3;
        return k1 + 16 * (qg * G + u) + 256 * q2;
    }
}

// inverse of bin_of for compile-time bins: the lane of a group and the result register that hold bin `bin`
struct BinSlot {
    int lane, reg;
};
template <int R3, int LGv = 16 * R3>
constexpr BinSlot slot_of_bin(int bin) {
    if (R3 == 1) return BinSlot{bin % LGv, bin / LGv};
    constexpr int G = 16 / R3;
    const int k1 = bin % 16, q1 = (bin / 16) % 16, q2 = bin / 256;
    const int qg = q1 / G, u = q1 % G;
    const int up = (u + ((k1 * R3 / 8) & (G - 1))) & (G - 1);
    return BinSlot{k1 * R3 + qg, up * R3 + q2};
}

// diagnostic builds only: timing-only ablations of the threshold-bit scan (MODE 6; wrong results): bit 0 no bit stores, bit 1 no chunk minima,
// bit 2 no staging of the per-bin thresholds, bit 3 no count of the cells over the absolute threshold
#ifndef RT_EXP6
#define RT_EXP6 0
#endif

// diagnostic builds only (tools/ablate.sh): stop the scan step after stage n, folding the live
// values into the row sums so nothing upstream is dead code.  0 = full kernel (the product).
#ifndef RT_ABLATE
#define RT_ABLATE 0
#endif


// diagnostic builds only (-DRT_STAMPS, tools/r3/stamps.sh): s_memtime stamps between the stages of the scan step; every
// wave adds up the cycles it spent in each stage (scalar registers) and leaves the sums in StftParams::dbg.  The
// stamps pin the instruction order (sched_barrier) and wait for the wave's LDS operations: the stage sums show where
// a wave's time goes, the kernel as a whole runs ~10 % slower than the product build.
#ifdef RT_STAMPS
#define RT_STAMP(k)                                                         \
    do {                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                  \
        const uint32_t now_ = (uint32_t)__builtin_amdgcn_s_memtime();       \
        st_acc[k] += now_ - st_prev;                                        \
        st_prev = now_;                                                     \
        __builtin_amdgcn_sched_barrier(0);                                  \
    } while (0)
constexpr int kStamps = 16;  // 0 .. 10 stage sums, 11 steps, 12 / 13 shader-clock and 100-MHz ticks over the step loop, 14 / 15 the 100-MHz clock at the wave's start and end
#else
#define RT_STAMP(k)
#endif

#define RT_ABLATE_STOP(n)                                               \
    if constexpr (RT_ABLATE == (n)) {                                   \
        _Pragma("unroll") for (int q_ = 0; q_ < 16; ++q_) acc[q_] += v[q_].x + v[q_].y; \
        continue;                                                       \
    }

// raw sample as it sits in HBM: complex64, or the RTL-SDR wire format (interleaved uint8 I, Q)
struct iq_u8 {
    uint16_t iq;  // low byte I, high byte Q
};

// IQ is read exactly once: the loads carry the non-temporal hint (load-only instantiation +4..9 %)
__device__ __forceinline__ cf load_iq(const cf *p) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v = __builtin_nontemporal_load(reinterpret_cast<const f2 *>(p));
    return cf{v.x, v.y};
}
__device__ __forceinline__ iq_u8 load_iq(const iq_u8 *p) { return iq_u8{__builtin_nontemporal_load(&p->iq)}; }

// Buffer loads (wave-uniform descriptor in SGPRs + one 32-bit lane offset + a scalar offset): the 16 loads of a
// segment share ONE address VGPR.  With flat addresses hipcc keeps a 64-bit pointer per 4 KiB of immediate range
// -- at nperseg 4096 eight register pairs recomputed in every step plus sixteen for the window, which is what
// pushed that kernel into scratch spills.  Out-of-range lanes read zero (num_records), so no index clamping.
typedef int rsrc_t __attribute__((ext_vector_type(4)));
typedef float buf_f2 __attribute__((ext_vector_type(2)));
__device__ buf_f2 raw_buffer_load_f2(rsrc_t rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2f32");
__device__ float raw_buffer_load_f1(rsrc_t rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");
typedef float buf_f4 __attribute__((ext_vector_type(4)));
__device__ buf_f4 raw_buffer_load_f4(rsrc_t rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ short raw_buffer_load_i16(rsrc_t rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.i16");

// `base` must be wave-uniform (the descriptor lives in SGPRs)
__device__ __forceinline__ rsrc_t make_rsrc(const void *base, uint32_t bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    rsrc_t r;
    r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    r.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));  // stride 0, no swizzle
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;  // gfx9 raw buffer: DATA_FORMAT 32
    return r;
}
constexpr int kAuxNT = 2;  // non-temporal
__device__ __forceinline__ cf buf_load_iq(rsrc_t r, int voff, int soff, cf) {
    const buf_f2 v = raw_buffer_load_f2(r, voff, soff, kAuxNT);
    return cf{v.x, v.y};
}
__device__ __forceinline__ iq_u8 buf_load_iq(rsrc_t r, int voff, int soff, iq_u8) {
    return iq_u8{(uint16_t)raw_buffer_load_i16(r, voff, soff, kAuxNT)};
}

// pyrtlsdr's packed_bytes_to_iq is (byte / 127.5) - 1 per component (in float64); here one
// float32 fma per component, at most one float32 ulp away, then float32 like complex64 input
__device__ __forceinline__ cf to_cf(cf x) { return x; }
__device__ __forceinline__ cf to_cf(iq_u8 x) {
    constexpr float c = 1.0f / 127.5f;
    return cf{__builtin_fmaf((float)(x.iq & 0xFFu), c, -1.0f), __builtin_fmaf((float)(x.iq >> 8), c, -1.0f)};
}

// stft_scan<.., QS>: the PER = 16 / QS consecutive samples a lane holds of one sixteenth of its segment, as 16-byte loads (complex64:
// two samples each; uint8 I/Q: 4 / 8 / 16 bytes in one load) into the registers e QS + m, e < PER.  The stream bases the host passes
// are aligned to these loads (rt_analyze.hip: process_impl); segments are 16 QS samples long, so every run is.
template <int PER, int QS>
__device__ __forceinline__ void load_iq_run(const cf *p, cf (&dst)[16], int m) {
    typedef float f4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int h = 0; h < PER / 2; ++h) {
        const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(p + 2 * h));
        dst[(2 * h) * QS + m] = cf{v.x, v.y};
        dst[(2 * h + 1) * QS + m] = cf{v.z, v.w};
    }
}
template <int PER, int QS>
__device__ __forceinline__ void load_iq_run(const iq_u8 *p, iq_u8 (&dst)[16], int m) {
    uint32_t w[PER / 2];
    if constexpr (PER == 2) {
        w[0] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(p));
    } else if constexpr (PER == 4) {
        typedef uint32_t u2 __attribute__((ext_vector_type(2)));
        const u2 v = __builtin_nontemporal_load(reinterpret_cast<const u2 *>(p));
        w[0] = v.x;  w[1] = v.y;
    } else {
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        const u4 v = __builtin_nontemporal_load(reinterpret_cast<const u4 *>(p));
        w[0] = v.x;  w[1] = v.y;  w[2] = v.z;  w[3] = v.w;
    }
#pragma unroll
    for (int h = 0; h < PER / 2; ++h) {
        dst[(2 * h) * QS + m] = iq_u8{(uint16_t)(w[h] & 0xFFFFu)};
        dst[(2 * h + 1) * QS + m] = iq_u8{(uint16_t)(w[h] >> 16)};
    }
}

// Threads per workgroup of the scan: 256.  One wave per workgroup at nperseg 256 (its lane groups never meet a workgroup
// barrier inside the step loop, and with four waves to a workgroup every wave waits at the end of an item for the slowest
// of the four: 12.7 us of a 109-us item, profiles/r03_g_wave_skew.txt) was built and measured: the scan launch 4 - 8 %
// shorter, but one partial row of sums per wave instead of per workgroup (63 per stream at config 2) makes every bucket
// wave of the detection wait for four times the loads: config 2 two lanes 742 k -> 661 k MS/s, one lane +-0, config 4 -1.5 %
// (profiles/r03_h_persistent_ab.txt).
#ifndef RT_ONE_WAVE_MAX_R3  // measured and dropped, kept as a switch: -DRT_ONE_WAVE_MAX_R3=1 = one wave per workgroup at nperseg 256
#define RT_ONE_WAVE_MAX_R3 0
#endif
// Which scans run their items on a chip-filling grid of persistent workgroups (stft_scan: "Work items"): nperseg >= 1024,
// every mode but the selective pass.  -DRT_NO_PERSIST=1: none (diagnostic).
#ifndef RT_NO_PERSIST
#define RT_NO_PERSIST 0
#endif
__device__ __forceinline__ int g_of(unsigned tid, int lg) { return (int)tid / lg; }
__host__ __device__ constexpr bool scan_persistent(int R3, int mode) { return mode != 5 && mode != 7 && !RT_NO_PERSIST && R3 >= 4; }
__host__ __device__ constexpr int scan_block(int R3) { return R3 <= RT_ONE_WAVE_MAX_R3 ? 64 : kBlock; }

// LIN: constant detrend by linearity.  FFT(w (x - m)) = FFT(w x) - m W with W = FFT(w); for a cosine-sum window of
// order <= 1 (hamming, hann, boxcar -- anything get_window() makes of them) W is real and zero outside bins 0 and +-1,
// so "subtract the mean from every sample" (32 subtractions per lane and step, and the transform waiting for the
// group-wide sum) becomes "subtract sum * W[k]/N from three output bins" (six fused multiply-adds, after pass 3).
// The host picks LIN when the window qualifies (rt_create); other windows keep the subtract-first form.
// Experiment (diagnostic builds, -DRT_EXP_DMA1024=1; round 6, the round-5 review's item 8): the complex64 kernels of nperseg 1024 take the
// next segment through an LDS landing zone of 8 KiB per wave (eight `buffer_load_dwordx4 ... lds` instead of sixteen register loads, issued
// as soon as the step's samples are out of the zone -- a whole transform ahead), at TWO workgroups per CU (the zones do not fit three
// times) with up to 256 registers.  Measured against the product on one box: EXPERIMENTS.md, round 6.
#ifndef RT_EXP_DMA1024
#define RT_EXP_DMA1024 0
#endif
__host__ __device__ constexpr bool scan_dma(int r3, bool u8, int qs) { return RT_EXP_DMA1024 && r3 == 4 && !u8 && qs == 0; }
__device__ void raw_buffer_load_lds_fwd(rsrc_t rsrc, __attribute__((address_space(3))) void *lds, int size, int voffset, int soffset, int offset, int aux)
    __asm("llvm.amdgcn.raw.buffer.load.lds");
// QS (round 6): nperseg 128 / 64 / 32 = 16 QS, lane groups of QS = 8 / 4 / 2 lanes (R3 = 1 in every other respect: one wave-private
// exchange, no barrier in the step loop, bin = lane + LG * register).  The transform is the 16 x QS form with the SMALL pass first:
// lane a of a group holds the 16 / QS consecutive samples n' = (16 / QS) a + e of every sixteenth of the segment, x[n' + 16 m'] --
// whole 128-byte lines per lane group and load instruction, eight 16-byte loads per lane and segment --
//   pass 1 (in-lane over m', 16 / QS transforms of QS points)   A[n'][k1] = sum_m' x[n' + 16 m'] W_QS^(m' k1),  times W_N^(n' k1)
//   exchange (LDS: row k1 of the group, column e QS + a)          lane k1 takes the sixteen n'
//   pass 2 (in-lane over n', one 16-point transform)            X[k1 + QS k2] = sum_n' A[n'][k1] W16^(n' k2)
// so everything behind the transform -- power, row sums, look-back tail, threshold bits, candidate emission, every MODE -- is the
// nperseg-256 code with LG = QS.  (Round 5 served these sizes by a kernel of its own on the dense path, stft_small: 282 k MS/s at
// nperseg 128 against the 650 k+ of the sparse path at 256.)
template <int R3, int MODE, bool U8 = false, bool LIN = false, int QS = 0>
// Experiment switch (default off): -DRT_WG4_MAX_R3=1 runs nperseg 256 at four workgroups per CU (its kernels need
// <= 124 VGPRs and, with 32 staged cells per wave, exactly 40 960 B of LDS).  Measured in round 2: one lane 0.792 ->
// 0.826 ms, two lanes 0.766 -> 0.788 ms per step (uint8 input +3 %): more waves do not help the complex64 scan.
#ifndef RT_WG4_MAX_R3
#define RT_WG4_MAX_R3 0
#endif
// Packed float32 butterflies (rt_fft.h: cfv) in the complex64 kernels of the R3 named by this mask (R3 is a power of two: bit R3;
// 0 = nowhere).  nperseg 256 (round 3: 164 VGPRs, no spill, +1.3 %) and 1024 (round 5: 168 VGPRs and 39 spilled ones, all of
// them lane constants of the rare tail-column block, +1 ... +2 % at BASELINE config 3 on two boxes, profiles/r05_l_*); NOT 512
// (12 spilled registers inside the step: -12 %) and not 2048 (spills).
#ifndef RT_PK_R3_MASK
#define RT_PK_R3_MASK (1 | 4)
#endif
// diagnostic builds only: -DRT_EXP_U8_PK=1 gives the uint8 kernels of nperseg 256 the packed butterflies too, at three workgroups per CU
// instead of four (the registers of the pairs) -- measured in round 5, EXPERIMENTS.md
#ifndef RT_EXP_U8_PK
#define RT_EXP_U8_PK 0
#endif
__global__ __launch_bounds__(scan_block(R3), scan_dma(R3, U8, QS) ? 2 : (R3 <= RT_WG4_MAX_R3 || (U8 && R3 == 1 && !RT_EXP_U8_PK)) ? 4 : 3)  // workgroups per CU = waves/SIMD: at most 128 / 168 VGPRs (left alone, hipcc takes 200 for nperseg 1024)
 void stft_scan(const StftParams p) {
    using raw_t = typename std::conditional<U8, iq_u8, cf>::type;
    static_assert(QS == 0 || (R3 == 1 && (QS == 2 || QS == 4 || QS == 8)), "QS: lane groups of 2 / 4 / 8 lanes, R3 = 1");
    constexpr int N = QS ? 16 * QS : 256 * R3;
    constexpr int LG = QS ? QS : 16 * R3;
    constexpr int PER = QS ? 16 / QS : 1;  // QS: consecutive samples a lane holds of every sixteenth of the segment
    constexpr int BLK = scan_block(R3);  // threads per workgroup
    constexpr int GPW = BLK / LG;  // lane groups per workgroup
    constexpr int G = 16 / R3;
    // Exchange rows: ROW complex values per lane, GPAD more per lane group.  QS (tools/lds_banks.py rules, `small` section): the
    // writes of a step go to column e QS + a of row k1 -- the lanes of a group side by side, the groups of a 16-lane write group in
    // different bank quarters by the pad -- and stay conflict-free; the ds_read_b128 of a lane's row are conflict-free at QS = 2, 2-way at 4 / 8.
    constexpr int ROW = (QS == 2) ? 16 : kRowF2;
    constexpr int GPAD = (U8 && QS == 8) ? 0 : QS;  // (uint8 at nperseg 128: with the pad the block is 512 B over a quarter of a CU's LDS, the four-workgroup form's limit; without it the writes are 2-way)

    // One LDS block carved by hand: at nperseg 256 the pieces add up to exactly 40 960 B, a quarter of a CU's LDS
    // (separate __shared__ arrays cannot have size zero, and their placeholders cost the fourth workgroup).
    constexpr bool W_IN_LDS = (R3 <= 8);  // N = 4096: the window comes from L2 as well (3 workgroups per CU)
    constexpr bool T1_IN_LDS = (R3 <= 4) && !scan_dma(R3, U8, QS);  // (the landing-zone experiment: the factored form, 6 KiB less -- two workgroups' 78 KiB fit a CU)
    constexpr bool T1_FACTORED = !T1_IN_LDS;
    // (uint8 input at nperseg 256 does run at four workgroups per CU: 106 VGPRs, +3 %)
    // (nperseg 2048: 96 -- with 128 the block is 54 576 B, and LDS is handed out in 512-byte pieces: three workgroups
    // would need 164 352 of the CU's 163 840 B, so the kernel ran at two; profiles/r03_d_stage_stamps.txt)
    constexpr bool DMA = scan_dma(R3, U8, QS);
    constexpr int kStage = (R3 <= RT_WG4_MAX_R3 || (U8 && R3 == 1 && !RT_EXP_U8_PK)) ? 32 : (R3 == 8) ? 96 : DMA ? 64 : kStageCap;  // candidate cells staged per wave before a flush
    constexpr size_t kXchB = sizeof(cf) * (BLK * ROW + GPW * GPAD);
    constexpr size_t kRedB = (LG > 64) ? sizeof(cf) * (BLK / 64) + 16 : 0;  // + the three tail_any words
    constexpr size_t kWB = W_IN_LDS ? sizeof(float4) * 4 * LG : 0;
    constexpr size_t kT1fB = T1_FACTORED ? sizeof(float4) * 2 * LG : 0;
    constexpr size_t kT1B = T1_IN_LDS ? sizeof(float4) * 8 * LG : 0;
    constexpr size_t kT2B = (R3 > 1) ? sizeof(float4) * 8 * R3 : 0;
    constexpr size_t kStageB = (MODE == 0 || MODE == 5 || MODE == 7) ? sizeof(uint2) * (BLK / 64) * kStage : 0;
    // MODE 6: the stream's per-bin second thresholds, staged per item in the order the lanes read them ([q][lane] float4 = the lane's
    // registers 4 q .. 4 q + 3).  Read from L2 in every step they queued behind the next segment's loads (vector-memory operations
    // return in order): the threshold test of a step waited for the prefetch of the next -- 2.11 ms for a scan whose bytes take 1.7
    // (profiles/r04_a_*).  Up to nperseg 1024, where the table fits beside the rest at three workgroups per CU.
    constexpr bool THR_LDS = (MODE == 6) && R3 <= 4;
    constexpr size_t kThrB = THR_LDS ? sizeof(float) * N : 0;
    // MODE 6: the threshold bits of four steps are collected per lane group in LDS and leave as ONE 8-byte store per lane (whole
    // 128-byte lines) instead of a 2-byte store per lane and step: the short stores cost the scan 10 % -- a vector-memory instruction
    // per step in a kernel whose waves queue for the address path (profiles/r05_c_mode6_ablations.txt: 1 760 -> 1 587 us without
    // them).  Where a group lives inside one wave and the block still fits three times into a CU's LDS: nperseg <= 512.
    constexpr bool BITS_LDS = (MODE == 6) && R3 <= 2 && (QS == 0 || QS >= 4);  // (a lane stores the words of four lanes: groups of two keep the short stores)
    constexpr size_t kBitsB = BITS_LDS ? sizeof(uint16_t) * 4 * BLK : 0;
    constexpr size_t kDmaB = DMA ? sizeof(cf) * N * (BLK / 64) : 0;  // a segment per wave
    __shared__ __attribute__((aligned(16))) unsigned char lds_block[kXchB + kRedB + kWB + kT1fB + kT1B + kT2B + kStageB + kThrB + kBitsB + kDmaB];
    // (wave index in a scalar register: an LDS-DMA takes its destination from M0)
    cf *const landing = reinterpret_cast<cf *>(lds_block + kXchB + kRedB + kWB + kT1fB + kT1B + kT2B + kStageB + kThrB + kBitsB) + (DMA ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * N : 0);
    uint16_t *const bits_lds = reinterpret_cast<uint16_t *>(lds_block + kXchB + kRedB + kWB + kT1fB + kT1B + kT2B + kStageB + kThrB) + (BITS_LDS ? g_of(threadIdx.x, LG) * 4 * LG : 0);  // this group's [4 steps][LG]
    float4 *const thr_lds = reinterpret_cast<float4 *>(lds_block + kXchB + kRedB + kWB + kT1fB + kT1B + kT2B + kStageB);
    cf *const xch = reinterpret_cast<cf *>(lds_block);
    cf *const red = reinterpret_cast<cf *>(lds_block + kXchB);
    uint32_t *const tail_any = reinterpret_cast<uint32_t *>(lds_block + kXchB + sizeof(cf) * (BLK / 64));  // LG > 64, see the tail columns below

    const int tid = threadIdx.x;
    const int g = tid / LG;
    const int lt = tid % LG;
    const int T = p.n_seg;
    const int L = p.segs_per_chunk;

    // the transform's arithmetic form: packed pairs where the registers are free (rt_fft.h), else scalar -- same results
    constexpr bool PK = ((RT_PK_R3_MASK & R3) != 0) && (!U8 || (RT_EXP_U8_PK && R3 == 1));
    using C = typename std::conditional<PK, cfv, cf>::type;
    C *gx = reinterpret_cast<C *>(xch + g * (LG * ROW + GPAD));  // this group's exchange rows

    // window and pass twiddles staged in LDS, laid out in the order the lanes
    // read them (16-byte pieces, consecutive lanes -> consecutive pieces), and
    // read just in time: keeping them in VGPRs would cost 62 registers per lane
    // and a wave per SIMD of occupancy.
    float4 *const w_lds = reinterpret_cast<float4 *>(lds_block + kXchB + kRedB);    // [m/4][lane]: w[lane + LG*(4*(m/4) + 0..3)]
    // N >= 2048: the full table (128 B per lane) does not fit LDS next to the exchange rows at 3 workgroups
    // per CU, and read from L2 it doubles the step's L1 fill traffic (measured +0.2 .. +0.34 ms per launch).
    // Only W^(a), W^(2a), W^(4a), W^(8a) are staged (32 B per lane); the other eleven factors are products
    // of two to four of them (W^(a k) with k in binary), 44 more VALU operations per step.
    float4 *const t1f_lds = reinterpret_cast<float4 *>(lds_block + kXchB + kRedB + kWB);  // [0][lane] = (W^a, W^2a), [1][lane] = (W^4a, W^8a)
    float4 *const t1_lds = reinterpret_cast<float4 *>(lds_block + kXchB + kRedB + kWB + kT1fB);   // [k/2][lane]: (tw1[lane][2*(k/2)], tw1[lane][2*(k/2)+1])
    float4 *const t2_lds = reinterpret_cast<float4 *>(lds_block + kXchB + kRedB + kWB + kT1fB + kT1B);  // [q/2][b]
    if constexpr (W_IN_LDS) {
        for (int idx = tid; idx < 4 * LG; idx += BLK) {
            const int mm = idx / LG, l = idx % LG;
            if constexpr (QS) {
                // register r = e QS + m' holds sample PER l + e + 16 m'
                auto smp = [&](int r) { return p.window[PER * l + r / QS + 16 * (r % QS)]; };
                w_lds[idx] = make_float4(smp(4 * mm), smp(4 * mm + 1), smp(4 * mm + 2), smp(4 * mm + 3));
            } else {
                w_lds[idx] = make_float4(p.window[l + LG * (4 * mm)], p.window[l + LG * (4 * mm + 1)],
                                         p.window[l + LG * (4 * mm + 2)], p.window[l + LG * (4 * mm + 3)]);
            }
        }
    }
    if constexpr (T1_IN_LDS) {
        for (int idx = tid; idx < 8 * LG; idx += BLK) {
            const int kk = idx / LG, l = idx % LG;
            const cf a = p.tw1[l * 16 + 2 * kk], b = p.tw1[l * 16 + 2 * kk + 1];
            t1_lds[idx] = make_float4(a.x, a.y, b.x, b.y);
        }
    }
    if constexpr (T1_FACTORED) {
        for (int l = tid; l < LG; l += BLK) {
            const cf w1 = p.tw1[l * 16 + 1], w2 = p.tw1[l * 16 + 2], w4 = p.tw1[l * 16 + 4], w8 = p.tw1[l * 16 + 8];
            t1f_lds[l] = make_float4(w1.x, w1.y, w2.x, w2.y);
            t1f_lds[LG + l] = make_float4(w4.x, w4.y, w8.x, w8.y);
        }
    }
    if constexpr (R3 > 1) {
        for (int idx = tid; idx < 8 * R3; idx += BLK) {
            const int kk = idx / R3, b = idx % R3;
            const cf x = p.tw2[b * 16 + 2 * kk], y = p.tw2[b * 16 + 2 * kk + 1];
            t2_lds[idx] = make_float4(x.x, x.y, y.x, y.y);
        }
    }
    // Work items.  An item is what a workgroup of the plain launch did: the chunks cb * GPW .. of one stream.  The
    // launch has just enough workgroups to fill the chip (rt_analyze.hip: scan_grid) and every workgroup keeps taking
    // items from a counter until none is left -- the tables above are staged once, and no slot waits for the
    // dispatcher: with one workgroup per item the hardware queue kept 2.2 - 2.7 of the 3 wave slots per SIMD filled
    // (it deals workgroups to the XCDs in strict rotation, so one full XCD holds back the others;
    // profiles/r03_f_wave_residency_timeline_*), at nperseg 256 a quarter of the launch.  Items are numbered latest
    // chunks first, across all streams: the ones that also write the look-back tail run longest and start first.
    // The first item is the workgroup's index; the ticket for the next one is drawn at the start of an item, so
    // its latency is covered.  MODE 5 keeps one workgroup per item (its items are listed per stream by plan_pass_b), and so
    // do nperseg 256 / 512 (scan_persistent): their launches at config 2 are 2.7 rounds of items per lane, and with two
    // lanes a chip-filling grid of long-lived workgroups shuts the other lane's kernels out until it ends (config 2, two
    // lanes: 740 k -> 676 k MS/s, one lane unchanged; a run-time choice per launch cost the nperseg-256 loop 5 %).
    constexpr bool PERSIST = scan_persistent(R3, MODE);
    uint32_t *const item_word = reinterpret_cast<uint32_t *>(lds_block + sizeof(cf) * 16);  // (padding of exchange row 0: never exchanged)
    const int n_items = p.n_streams * p.blocks_per_stream;
    int item = blockIdx.x;
#if defined(RT_EXP_PRIO) && RT_EXP_PRIO == 1  // experiment: a fixed issue priority per workgroup slot of a CU (the persistent grid is 3 x the CUs)
    if constexpr (PERSIST) {
        const unsigned slot = blockIdx.x / (gridDim.x / 3);
        if (slot == 0) __builtin_amdgcn_s_setprio(2);
        else if (slot == 1) __builtin_amdgcn_s_setprio(1);
    }
#endif
    __syncthreads();
  for (;;) {  // one item per round
#ifdef RT_STAMPS
    const uint32_t st_wave_start = (uint32_t)__builtin_amdgcn_s_memrealtime();  // (per item)
#endif
    uint32_t ticket = 0;
    if constexpr (PERSIST) {
        if (item >= n_items) break;  // (uniform: every thread of the workgroup holds the same item)
        if (tid == 0) ticket = atomicAdd(p.work, 1u);
    }
    const int s_pos = item % p.n_streams;
    const int s = p.stream_list ? p.stream_list[s_pos] : s_pos;
    if constexpr (LIN) {
        if (p.sub_first && p.sub_first[s] != 0) {  // a stream the guard has marked: the subtract-first launch behind this one analyses it
            if constexpr (!PERSIST) {
                return;
            } else {
                __syncthreads();
                if (tid == 0) *item_word = ticket;
                __syncthreads();
                item = (int)gridDim.x + (int)*item_word;
                continue;
            }
        }
    }
    const int cb = p.blocks_per_stream - 1 - item / p.n_streams;
    int chunk = cb * GPW + g;
    if constexpr (MODE == 5) {
        // pass B of the run-length pre-filter: the stream's workgroups take the chunks plan_pass_b packed for them
        const int pb = item / p.n_streams;
        if (pb >= p.item_count[s]) return;
        chunk = p.item_chunks[((int64_t)s * p.blocks_per_stream + pb) * GPW + g];
    }
    // MODE 7: the item is (up to) GPW * L entries of the stream's segment list, dealt to the workgroup's lane groups in equal
    // runs of `per7` consecutive entries, and the step loop ends after per7 steps.  (L entries to a lane group and L steps whatever
    // the list held -- the first version -- left 12 of 16 lane groups idle at the reference's default geometry, where a stream
    // lists 137 of its 1 171 segments, and walked 32 steps for 9: 0.58 ms for an eighth of the samples, profiles/r04_a_*.)
    int e0 = 0, n_mine = 0, per7 = 0;
    if constexpr (MODE == 7) {
        const int pb = item / p.n_streams;
        const int cnt = p.seg_count[s];
        const int first = pb * GPW * p.segs_per_chunk;
        if (first >= cnt) return;  // (workgroup-uniform)
        const int n_item = cnt - first < GPW * p.segs_per_chunk ? cnt - first : GPW * p.segs_per_chunk;
        per7 = (n_item + GPW - 1) / GPW;
        e0 = first + g * per7;
        n_mine = n_item - g * per7;
        n_mine = n_mine < 0 ? 0 : (n_mine > per7 ? per7 : n_mine);
    }
    const bool chunk_ok = (MODE == 7) ? (n_mine > 0) : (chunk < p.chunks);
    const int c0 = chunk * p.segs_per_chunk;
    if constexpr (THR_LDS) {
        if (p.thr_bin && !(RT_EXP6 & 4)) {  // (the previous item's readers are behind the barrier that ended it; this item's first step is behind the next one)
            // max(absolute threshold, the bin's own): `!(P < a) && !(P < b)` is `!(P < max(a, b))`, one test per cell instead of two
            const float thr_abs = p.thr_s ? p.thr_s[s] : p.thr;
            const float4 *src = reinterpret_cast<const float4 *>(p.thr_bin + (int64_t)s * N);  // [lane][q]
            for (int idx = tid; idx < 4 * LG; idx += BLK) {
                const float4 t4 = src[(idx % LG) * 4 + idx / LG];
                thr_lds[idx] = make_float4(fmaxf(t4.x, thr_abs), fmaxf(t4.y, thr_abs), fmaxf(t4.z, thr_abs), fmaxf(t4.w, thr_abs));
            }
        }
        __syncthreads();
    }
    // MODE 6 with staged thresholds: the bits of the absolute threshold alone (`hot`) are only built where something reads them --
    // the chunk bits (p.full), the sparse tail's masks (items that reach into the last tail_cols segments) -- and the count for
    // AUTO's probes (abs_hot) comes from every eighth segment elsewhere: 48 of ~630 vector instructions of every step at the
    // reference's default geometry, where the noise passes the absolute threshold in nearly every lane.
    bool item_hot = true;  // wave-uniform
    const int samp_period = abs_sample_period(L);  // (see sampled_abs below)
    const int samp_phase = __builtin_amdgcn_readfirstlane(cb + (int)(threadIdx.x >> 6));
    if constexpr (THR_LDS) {
        if (p.thr_bin) item_hot = __builtin_amdgcn_ballot_w64(p.full != nullptr || c0 + p.segs_per_chunk > p.n_seg - p.tail_cols) != 0ull;
    }
    if constexpr (LG > 64) {
        // (behind the barrier that ended the previous item, ahead of this item's first one: group_sum / the rows barrier)
        if (tid < 3) tail_any[tid] = (tid == 1) ? 1u : 0u;  // the first step (i = 1) writes its whole column
    }

    float acc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float thr = p.thr_s ? p.thr_s[s] : p.thr;  // wave-uniform
    // LIN: the (lane, register) slots of bins 0, 1 and N-1, slots in one register merged: lin_k[j] is this lane's
    // coefficient for register kLinReg[j] (zero in all but three lanes of a group)
    constexpr BinSlot kS0 = slot_of_bin<R3, LG>(0), kS1 = slot_of_bin<R3, LG>(1), kS2 = slot_of_bin<R3, LG>(N - 1);
    constexpr int kLinIdx1 = (kS1.reg == kS0.reg) ? 0 : 1;
    constexpr int kLinIdx2 = (kS2.reg == kS0.reg) ? 0 : (kS2.reg == kS1.reg) ? kLinIdx1 : kLinIdx1 + 1;
    constexpr int kLinRegs = kLinIdx2 > kLinIdx1 ? kLinIdx2 + 1 : kLinIdx1 + 1;
    constexpr int kLinReg[3] = {kS0.reg, kLinIdx1 == 1 ? kS1.reg : kS2.reg, kS2.reg};
    float lin_k[3] = {0.f, 0.f, 0.f};
    float dc_acc = 0.f;  // LIN: sum over the item's segments of |sum of the segment's samples|^2 (the same in every lane of a group)
    if constexpr (LIN) {
        if (lt == kS0.lane) lin_k[0] += p.lin_c[0];
        if (lt == kS1.lane) lin_k[kLinIdx1] += p.lin_c[1];
        if (lt == kS2.lane) lin_k[kLinIdx2] += p.lin_c[2];
    }
    uint32_t next_hot = 0;  // hot bits of the segment one later in time (MODE 0)
    // candidate staging: 128 cells per wave, flushed with one returned atomic per bucket
    uint2 *const stage = reinterpret_cast<uint2 *>(lds_block + kXchB + kRedB + kWB + kT1fB + kT1B + kT2B);
    uint2 *stg = stage + ((MODE == 0 || MODE == 5 || MODE == 7) ? (tid >> 6) * kStage : 0);  // this wave's staging area
    constexpr int kStageLimit = kStage;
    int stg_n = 0;                                                    // wave-uniform fill level
    bool gave_up = false;  // wave-uniform: a candidate list of this stream has overflowed, the call will be re-run dense

    const raw_t *stream_iq = reinterpret_cast<const raw_t *>(p.iq) + (int64_t)s * p.stream_stride;
    constexpr bool EMIT = (MODE == 0 || MODE == 5 || MODE == 7);   // candidate cells go to the bucket lists
    constexpr bool FLAGS = (MODE == 0 || MODE == 4 || MODE == 6);  // threshold bits are kept (chunk bits of the run-length pre-filter; MODE 6: every cell's)
    constexpr bool SUMS = (MODE != 2 && MODE != 5 && MODE != 7);   // row sums and look-back tail (MODE 5 / 7 repeat segments of a scan that wrote them)
    constexpr bool LISTED = (MODE == 7);  // the steps take the segments plan_runs listed, not a chunk's
    // Steps 1 .. L walk the chunk down from its latest segment.  A cell is a candidate cell if it passes the threshold or
    // directly precedes one that does (T11); for the chunk's lowest segment c0 that concerns a cell of the neighbour
    // below, whose owner cannot know.  So where (and only where) a lowest cell is hot, the wave (workgroup, for lane
    // groups of several waves) takes step L + 1 on segment c0 - 1 and emits the cells there that precede a hot one and
    // are not hot themselves (those the owner emits): ~10 % of the chunks on sparse input, where a halo segment read
    // and transformed by every chunk cost 1/33 of all loads and arithmetic.
    // nperseg 4096 keeps that halo segment (BELOW = false: step 0 on segment c0 + L, whose only product is next_hot):
    // there the extra workgroup barrier and registers of the conditional step cost more than the halo (+2.5 %).
#ifndef RT_BELOW_MAX_R3
#define RT_BELOW_MAX_R3 8
#endif
    constexpr bool BELOW = (R3 <= RT_BELOW_MAX_R3);
    const int i_first = (!BELOW && EMIT && !LISTED) ? 0 : 1;
    int n_steps = LISTED ? per7 : L;

    uint32_t allhot = 0xFFFFu;  // FLAGS: the chunk's bits so far
    uint32_t n_abs = 0;         // MODE 4 / 6: this lane's cells at or above the absolute threshold (StftParams::abs_hot)
    uint32_t need = 0xFFFFu;                      // MODE 5: the lane's bins that may emit in every segment of the chunk
    uint32_t first_nxt = 0u;                      // MODE 5, chunk 0: the bins whose run through t = 0 reaches the requested segment
    bool group_need = true;                       // MODE 5: this lane group transforms its chunk
    if constexpr (MODE == 5) {
        need = 0u;
        uint32_t need_run = 0u;
        if (chunk_ok) {
            const uint16_t *f = p.full + ((int64_t)s * p.chunks + chunk) * LG + lt;
            need = f[0];
            if (chunk > 0) need |= f[-LG];
            if (chunk + 1 < p.chunks) need |= f[LG];
            need_run = need;
            if (chunk == 0) need_run |= p.first[(int64_t)s * p.segs_per_chunk * LG + lt];  // any run through t = 0
        }
        if constexpr (LG > 64) {
            // the step loop has workgroup barriers: the whole workgroup (1 or 2 chunks) goes or stays
            if (!__syncthreads_or(need_run != 0u)) return;
        } else {
            const unsigned long long any = __builtin_amdgcn_ballot_w64(need_run != 0u);
            if (any == 0ull) return;  // nothing below needs this wave (no workgroup barrier follows in MODE 5)
            // the wave runs, but the lane groups whose chunk is not needed request no memory and emit nothing
            if constexpr (LG < 64) group_need = ((any >> ((threadIdx.x & 63) & ~(LG - 1))) & ((1ull << LG) - 1ull)) != 0ull;
        }
    }

    if constexpr (LISTED) {
        need = 0u;  // (what a step emits comes with its segment: first_nxt)
        if constexpr (LG <= 64) {
            const unsigned long long any = __builtin_amdgcn_ballot_w64(n_mine > 0);
            if (any == 0ull) return;  // (no workgroup barrier follows in MODE 7 either)
            if constexpr (LG < 64) group_need = n_mine > 0;
        }
    }
    int seg7_cur = -1, seg7_nxt = -1;  // LISTED: this step's segment and the next step's (-1: none)
    if constexpr (LISTED) {
        const int32_t *lst = p.seg_list + (int64_t)s * T + e0;
        if (n_mine > 0) seg7_cur = lst[0];
        if (n_mine > 1) seg7_nxt = lst[1];
    }

    // software pipeline: the 16 loads of the next segment are issued before the
    // current one is transformed, so their HBM latency hides under ~700 VALU ops.
    // Loads are unconditional (segment index clamped into the stream): lanes of
    // idle groups / past-the-end steps read valid memory and discard it.
    const int seg_hi = T - 1;
    raw_t nxt[16] = {};
    // LG >= 64: a lane group is one or more whole waves, so the segment index is wave-uniform and the loads go
    // through a buffer descriptor per segment (a past-the-end segment gets an empty one: its lanes read zeros,
    // which idle groups discard).  Smaller groups share a wave with other chunks: flat loads, indices clamped.
    constexpr bool BUF_LOADS = (LG >= 64);
    const int chunk_u = BUF_LOADS ? __builtin_amdgcn_readfirstlane(chunk) : 0;
    auto request_segment = [&](int seg_req) {
        if constexpr (MODE == 5) {
            first_nxt = 0u;
            if (chunk == 0 && chunk_ok && seg_req >= 0 && seg_req < L) first_nxt = p.first[((int64_t)s * L + seg_req) * LG + lt];
        }
        if constexpr (LISTED) {
            first_nxt = 0u;
            if (seg_req >= 0) first_nxt = p.cell_need[((int64_t)s * T + seg_req) * LG + lt];  // the cells of that segment to emit
        }
        if constexpr (BUF_LOADS) {
            const int sg = LISTED ? __builtin_amdgcn_readfirstlane(seg_req)
                                  : __builtin_amdgcn_readfirstlane(chunk_u * L + (seg_req - c0));  // == seg_req, in SGPRs
#ifdef RT_EXP_ALIAS
            const raw_t *base = reinterpret_cast<const raw_t *>(p.iq) + (int64_t)(sg & RT_EXP_ALIAS) * N;
            const rsrc_t r = make_rsrc(base, (uint32_t)(N * sizeof(raw_t)));
#else
            // (sg < 0: the step below the chunk of a workgroup whose other lane group holds chunk 0)
            const rsrc_t r = make_rsrc(stream_iq + (int64_t)(sg < 0 ? 0 : sg) * N, (sg >= 0 && sg < T) ? (uint32_t)(N * sizeof(raw_t)) : 0u);
#endif
            if constexpr (DMA) {
                // the wave's landing zone: piece j = 1 KiB of consecutive samples (a lane 16 bytes), eight instructions, no registers
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    raw_buffer_load_lds_fwd(r, (__attribute__((address_space(3))) void *)(reinterpret_cast<char *>(landing) + 1024 * j), 16, (int)(threadIdx.x & 63) * 16, 1024 * j, 0, kAuxNT);
            } else {
#pragma unroll
                for (int m = 0; m < 16; ++m) nxt[m] = buf_load_iq(r, lt * (int)sizeof(raw_t), LG * m * (int)sizeof(raw_t), raw_t{});
            }
        } else if ((MODE != 5 && MODE != 7) || group_need) {
            int sc = seg_req < seg_hi ? seg_req : seg_hi;
            sc = sc < 0 ? 0 : sc;  // (step L + 1 of a wave that also holds chunk 0)
#ifdef RT_EXP_ALIAS  // diagnostic build (tools/variant.sh alias -DRT_EXP_ALIAS=63): every load hits the same 64 segments of
                     // stream 0 (L2-resident) -- the scan kernel without HBM, i.e. its arithmetic + LDS floor; a mask of
                     // 8191 keeps a whole stream (16 MB at nperseg 256: misses L2, stays in the Infinity Cache)
            const raw_t *src = reinterpret_cast<const raw_t *>(p.iq) + (int64_t)(sc & RT_EXP_ALIAS) * N + PER * lt;
#else
            const raw_t *src = stream_iq + (int64_t)sc * N + PER * lt;
#endif
            if constexpr (QS) {
                // PER consecutive samples of every sixteenth of the segment: register e QS + m' <- sample PER lt + e + 16 m'
#pragma unroll
                for (int m = 0; m < QS; ++m) load_iq_run<PER, QS>(src + 16 * m, nxt, m);
            } else {
#pragma unroll
                for (int m = 0; m < 16; ++m) nxt[m] = load_iq(src + LG * m);
            }
        }
    };
    request_segment(LISTED ? seg7_cur : c0 + L - i_first);
    // MODE 6 (BITS_LDS): the threshold bits of steps last - rows + 1 .. last leave the group's LDS rows as one 8-byte store per lane.
    // In memory the block is segment c0 + L - last (the lowest) and up to three later ones, LG x 2 bytes each: lane j of the group
    // stores the four 16-bit words of lanes 4 (j % (LG / 4)) .. + 3 of row j / (LG / 4).  Called at the HEAD of a step, ahead of its
    // loads: vector-memory operations return in order and the step's samples are waited for with everything older, so a store
    // issued behind the loads (the first version: at the end of every step) made each step wait for its own store's
    // acknowledgement as well.
    auto flush_bits = [&](int last, int rows) {
        if constexpr (BITS_LDS && !(RT_EXP6 & 1)) {
            wave_sync();
            int lt_f = lt;
            asm volatile("" : "+v"(lt_f));
            const int m = lt_f / (LG / 4), qd = lt_f % (LG / 4);
            const int sg = c0 + L - last + m;
            if (chunk_ok && m < rows && sg >= 0 && sg < T) {
                const uint2 v = *reinterpret_cast<const uint2 *>(bits_lds + ((last - m - 1) & 3) * LG + 4 * qd);
                *reinterpret_cast<uint2 *>(p.cell_hot + ((int64_t)s * T + sg) * LG + 4 * qd) = v;
            }
            wave_sync();  // (the rows are rewritten from this step on)
        }
    };
#ifdef RT_STAMPS
    uint32_t st_acc[kStamps] = {};
    uint32_t st_prev = (uint32_t)__builtin_amdgcn_s_memtime();
    const uint32_t st_t0 = st_prev, st_r0 = (uint32_t)__builtin_amdgcn_s_memrealtime();
#endif

    for (int i = i_first; i <= n_steps; ++i) {
        RT_STAMP(0);  // loop control, the previous step's candidate test
        const int seg = LISTED ? seg7_cur : c0 + L - i;
        const bool halo = LISTED ? false : (BELOW ? (i > L) : (i == 0));  // the step below (above) the chunk: no sums, tail, chunk bits
        const bool active = chunk_ok && seg < T && seg >= 0;

        C v[16];
        if constexpr (DMA) {
            // the segment has landed (everything older than this point has: vmcnt(0)); its samples out of the zone, which is then free
            // for the next segment's pieces
            __builtin_amdgcn_s_waitcnt(0x0F70);
            __builtin_amdgcn_sched_barrier(0);
            int lt_d = lt;
            asm volatile("" : "+v"(lt_d));
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const cf x = landing[lt_d + LG * m];
                v[m] = make_c<C>(x.x, x.y);
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the reads have returned
            __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const cf x = to_cf(nxt[m]);
                v[m] = make_c<C>(x.x, x.y);
            }
        }
        const uint32_t need_seg = need | first_nxt;  // (first_nxt was requested for this step's segment)
        if constexpr (LISTED) {
            // consumed: the word goes back to zero (the planner writes only the words that keep anything: plan_runs).  Here, a step after
            // the request, its value has arrived anyway -- a store that depended on it at the request would have waited for it there
            if (first_nxt && seg >= 0) const_cast<uint16_t *>(p.cell_need)[((int64_t)s * T + seg) * LG + lt] = 0;
        }
        if constexpr (BITS_LDS) {
            if (i > 1 && ((i - 1) & 3) == 0) flush_bits(i - 1, 4);
        }
        // N = 4096: the window comes from L2.  Vector-memory operations return in order, so these loads must be
        // issued BEFORE the next segment's: waiting for them afterwards (`s_waitcnt vmcnt(0)`) would wait for the
        // whole prefetch, i.e. expose an HBM round trip in every step (it did: 1.01 ms per launch).
        float wreg[W_IN_LDS ? 1 : 16];
        if constexpr (!W_IN_LDS) {
#ifdef RT_EXP_NOWIN  // timing-only diagnostic: no window loads (wrong spectra)
#pragma unroll
            for (int m = 0; m < 16; ++m) wreg[m] = 0.5f + 0.01f * m;
#else
            // (a lane's sixteen coefficients lie side by side in the transposed table: four 16-byte loads instead of
            // sixteen dword loads -- a vector-memory instruction costs the issuing wave ~50 cycles behind the other
            // waves' HBM requests, L2 hit or not: profiles/r03_d_stage_stamps.txt, stage 1)
            const rsrc_t rw = make_rsrc(p.window_t, (uint32_t)(N * sizeof(float)));
#pragma unroll
            for (int mm = 0; mm < 4; ++mm) {
                const buf_f4 w4 = raw_buffer_load_f4(rw, lt * 64, mm * 16, 0);
                wreg[4 * mm] = w4.x;  wreg[4 * mm + 1] = w4.y;  wreg[4 * mm + 2] = w4.z;  wreg[4 * mm + 3] = w4.w;
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the order of the two groups of loads
#endif
        }
#if defined(RT_EXP_PRIO) && RT_EXP_PRIO == 2  // experiment: the wave that requests memory goes first
        if constexpr (PERSIST) __builtin_amdgcn_s_setprio(3);
#endif
        // next step's segment (the segment below the chunk is requested at the end of step L, once it is known to be needed)
        if constexpr (LISTED) {
            request_segment(seg7_nxt);
            seg7_cur = seg7_nxt;  // (for the next step; `seg` above is this step's)
            seg7_nxt = (i + 1 < n_mine) ? p.seg_list[(int64_t)s * T + e0 + i + 1] : -1;  // consumed a step later: its latency is covered
        } else if constexpr (BELOW) {
            if (i < L) request_segment(seg - 1);
        } else {
            request_segment((i < L) ? seg - 1 : seg);  // (the last step re-reads its own: harmless, keeps the loop uniform)
        }

#if defined(RT_EXP_PRIO) && RT_EXP_PRIO == 2
        if constexpr (PERSIST) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(0); }
#endif
        RT_STAMP(1);  // issue of the window loads (nperseg 4096) and the next segment's loads
        if constexpr (MODE == 3) {
            // traffic calibration: the scan's exact load stream, nothing else
#pragma unroll
            for (int m = 0; m < 16; ++m) acc[0] += v[m].x + v[m].y;
            continue;
        }

        // detrend='constant': subtract the segment mean (scipy _signaltools.py:3926)
        cf sum;
        {
            C s8[8], s4[4];
#pragma unroll
            for (int m = 0; m < 8; ++m) s8[m] = cadd(v[m], v[m + 8]);
#pragma unroll
            for (int m = 0; m < 4; ++m) s4[m] = cadd(s8[m], s8[m + 4]);
            const C s1 = cadd(cadd(s4[0], s4[2]), cadd(s4[1], s4[3]));
            sum = cf{s1.x, s1.y};
        }
        RT_STAMP(2);  // wait for this segment's samples (requested a step ago) + the in-lane sums
        if constexpr (LIN && LG > 64) {
            // The sum is only needed after pass 3.  This wave's share goes to `red` behind the barrier that frees the
            // exchange rows (every wave has read the previous step's shares by then: it read them right behind that
            // step's exchange barrier) and is read back behind this step's exchange barrier, where its latency hides
            // under two passes -- instead of a write / barrier / read chain at the head of the step.
            // (Measured and dropped: counters in LDS instead of this barrier -- a wave adds one when it is done with the
            // rows and waits only before its next stores into them.  The waves of a group reach the first barrier of a
            // step ~1 000 cycles apart; without this one they wait as long at the exchange barrier instead:
            // profiles/r03_d_stage_stamps.txt, profiles/r03_g_wave_skew.txt.)
            sum = wave_sum<LG>(sum);
            __syncthreads();  // the previous step's last use of the exchange rows is over in every wave
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
        } else {
            sum = group_sum<LG>(sum, red);  // (for LG > 64 its barrier also frees the exchange rows)
        }
        RT_STAMP(3);  // group sum (lane shuffles, nperseg >= 2048: workgroup barrier)
        const cf mean_s = LIN ? cf{0.f, 0.f} : cscale(sum, 1.0f / (float)N);
        const C mean = make_c<C>(mean_s.x, mean_s.y);
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) {
            float4 w4;
            if constexpr (W_IN_LDS) {
                w4 = w_lds[mm * LG + lt];
            } else {
                w4 = make_float4(wreg[4 * mm], wreg[4 * mm + 1], wreg[4 * mm + 2], wreg[4 * mm + 3]);
            }
            if constexpr (LIN) {
                v[4 * mm + 0] = cscale(v[4 * mm + 0], w4.x);
                v[4 * mm + 1] = cscale(v[4 * mm + 1], w4.y);
                v[4 * mm + 2] = cscale(v[4 * mm + 2], w4.z);
                v[4 * mm + 3] = cscale(v[4 * mm + 3], w4.w);
            } else {
                v[4 * mm + 0] = cscale(csub(v[4 * mm + 0], mean), w4.x);
                v[4 * mm + 1] = cscale(csub(v[4 * mm + 1], mean), w4.y);
                v[4 * mm + 2] = cscale(csub(v[4 * mm + 2], mean), w4.z);
                v[4 * mm + 3] = cscale(csub(v[4 * mm + 3], mean), w4.w);
            }
        }

        RT_STAMP(4);  // window multiply (nperseg 4096: waits for the window loads)
        RT_ABLATE_STOP(1)  // loads + detrend + window
        // pass 1
        if constexpr (QS) dft_groups<QS>(v); else dft16(v);  // (QS: v[e QS + k1] = A[PER lt + e][k1])
        if constexpr (T1_FACTORED) {
            const float4 ta = t1f_lds[lt], tb = t1f_lds[LG + lt];
            const C w1 = make_c<C>(ta.x, ta.y), w2 = make_c<C>(ta.z, ta.w), w4 = make_c<C>(tb.x, tb.y), w8 = make_c<C>(tb.z, tb.w);
            const C w3 = cmul(w1, w2), w5 = cmul(w4, w1), w6 = cmul(w4, w2), w7 = cmul(w4, w3);
            v[1] = cmul(v[1], w1);
            v[2] = cmul(v[2], w2);
            v[3] = cmul(v[3], w3);
            v[4] = cmul(v[4], w4);
            v[5] = cmul(v[5], w5);
            v[6] = cmul(v[6], w6);
            v[7] = cmul(v[7], w7);
            v[8] = cmul(v[8], w8);
            v[9] = cmul(v[9], cmul(w8, w1));
            v[10] = cmul(v[10], cmul(w8, w2));
            v[11] = cmul(v[11], cmul(w8, w3));
            v[12] = cmul(v[12], cmul(w8, w4));
            v[13] = cmul(v[13], cmul(w8, w5));
            v[14] = cmul(v[14], cmul(w8, w6));
            v[15] = cmul(v[15], cmul(w8, w7));
        } else {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const float4 t = t1_lds[kk * LG + lt];
                // (QS: register e QS + k1 takes W_N^((PER lt + e) k1) -- one for k1 = 0)
                if (kk && (!QS || (2 * kk) % (QS ? QS : 1) != 0)) v[2 * kk] = cmul(v[2 * kk], make_c<C>(t.x, t.y));
                if (!QS || (2 * kk + 1) % (QS ? QS : 1) != 0) v[2 * kk + 1] = cmul(v[2 * kk + 1], make_c<C>(t.z, t.w));
            }
        }

        RT_STAMP(5);  // pass 1 and its twiddles
        RT_ABLATE_STOP(2)  // + pass 1 and twiddles
        // exchange 1: element (a = lt, k1) -> row k1*R3 + b, column c
        if constexpr (QS) {
#pragma unroll
            for (int r = 0; r < 16; ++r) gx[(r % QS) * ROW + (r / QS) * QS + lt] = v[r];  // row k1, column e QS + lt
        } else {
            const int b = lt % R3, c = (lt / R3 + x1_rotation<R3>(lt % R3)) & 15;
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) gx[(k1 * R3 + b) * kRowF2 + c] = v[k1];
        }
        RT_STAMP(6);  // exchange 1: stores
        group_sync<LG>();
        RT_STAMP(7);  // exchange 1: barrier (nperseg >= 2048)
        if constexpr (LIN && LG > 64) {
            constexpr int WPG = LG / 64;
            const int w0 = ((threadIdx.x >> 6) / WPG) * WPG;
            cf t{0.f, 0.f};
#pragma unroll
            for (int w = 0; w < WPG; ++w) t = cadd(t, red[w0 + w]);
            sum = t;  // (used after pass 3)
        }
        if constexpr (QS) {
            // column c = e QS + a of the lane's row holds n' = PER a + e: back to the order of n' (a renaming of registers)
            const float4 *row = reinterpret_cast<const float4 *>(gx + lt * ROW);
            C col[16];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float4 q = row[j];
                col[2 * j] = make_c<C>(q.x, q.y);
                col[2 * j + 1] = make_c<C>(q.z, q.w);
            }
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) v[n1] = col[(n1 % PER) * QS + n1 / PER];
        } else {
            const float4 *row = reinterpret_cast<const float4 *>(gx + lt * kRowF2);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float4 q = row[j];
                v[2 * j] = make_c<C>(q.x, q.y);
                v[2 * j + 1] = make_c<C>(q.z, q.w);
            }
        }

        RT_ABLATE_STOP(3)  // + LDS exchange
        // pass 2
        dft16(v);
        RT_STAMP(8);  // exchange 1: loads, pass 2
        RT_ABLATE_STOP(4)  // + pass 2

        if constexpr (R3 > 1) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const float4 t = t2_lds[kk * R3 + (lt % R3)];
                if (kk) v[2 * kk] = cmul(v[2 * kk], make_c<C>(t.x, t.y));
                v[2 * kk + 1] = cmul(v[2 * kk + 1], make_c<C>(t.z, t.w));
            }
            // Exchange 2 stays inside the R3 lanes that share k1 (R3 consecutive lanes, R3 consecutive
            // rows: the rows those very lanes read in exchange 1), so wave-level ordering is enough.
            wave_sync();
            {
                const int k1 = lt / R3, b = lt % R3, sh = x2_rotation<R3>(lt / R3);
#pragma unroll
                for (int q1 = 0; q1 < 16; ++q1)
                    gx[(k1 * R3 + q1 / G) * kRowF2 + ((q1 % G + sh) & (G - 1)) * R3 + b] = v[q1];
            }
            wave_sync();
            {
                const float4 *row = reinterpret_cast<const float4 *>(gx + lt * kRowF2);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float4 q = row[j];
                    v[2 * j] = make_c<C>(q.x, q.y);
                    v[2 * j + 1] = make_c<C>(q.z, q.w);
                }
            }
            RT_ABLATE_STOP(5)  // + pass-2 twiddles and exchange 2
            // pass 3
            dft_groups<R3>(v);
        }
        // rows are free for the next segment: inside a wave by program order; across the waves of a larger
        // group the next step's mean reduction has the barrier (group_sum)
        wave_sync();
        RT_STAMP(9);  // pass-2 twiddles, exchange 2, pass 3
        RT_ABLATE_STOP(6)  // + pass 3

        if constexpr (LIN) {
            if constexpr (SUMS) {
                if (active && !halo) dc_acc = __builtin_fmaf(sum.x, sum.x, __builtin_fmaf(sum.y, sum.y, dc_acc));  // (guard of this form: StftParams::dc_flag)
            }
            // X[k] -= (sum x) * W[k]/N for k in {0, 1, N-1}: the constant detrend, applied to the transform
#pragma unroll
            for (int j = 0; j < kLinRegs; ++j) {
                v[kLinReg[j]].x = __builtin_fmaf(-lin_k[j], sum.x, v[kLinReg[j]].x);
                v[kLinReg[j]].y = __builtin_fmaf(-lin_k[j], sum.y, v[kLinReg[j]].y);
            }
        }
        // |X|^2 * scale  (scipy _spectral_py.py:2126-2128); sqrt(scale) is folded into the window table
        float P[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) P[r] = __builtin_fmaf(v[r].x, v[r].x, v[r].y * v[r].y);

        if constexpr (SUMS) {
            if (active && !halo) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] += P[r];
            }
        }
        if constexpr (RT_ABLATE == 8) continue;  // + power and row sums only
        {
            // spectrogram row (dense modes) and look-back tail column (last K segments)
            const int col = seg - (T - p.tail_cols);
            const bool to_spec = (MODE == 1 || MODE == 2) && active && !halo;
            const bool to_tail = SUMS && active && !halo && col >= 0;
            float *spec_dst = p.spec + ((int64_t)(p.spec_by_stream ? s : s_pos) * T + seg) * N;
            float *tail_dst = p.tail + ((int64_t)s * p.tail_cols + col) * N;
            // Sparse tail (the scans that keep threshold bits, MODE 0 / 4): the next buffer's look-back walks down from
            // the last segment while the cells pass the absolute threshold and stops ON the first that does not
            // (rt_core.h: walk_start; cell_above implies >= thr), so it can only reach a cell whose later cells are all
            // hot.  A cell is written iff the later cells OF ITS CHUNK are -- `allhot` before this step's update, all
            // ones at the chunk's last segment: a superset (a walk that crosses into a chunk enters at its last
            // segment), one column per chunk instead of 32 on sparse input.  Cells not written keep stale values
            // that no walk reaches.  The dense scan (MODE 1) writes every cell.
            const uint32_t tail_mask = FLAGS ? allhot : 0xFFFFu;
            if constexpr (R3 == 1) {
                // bin = lane + 16 r: every store instruction already writes 64-byte runs
                if (to_spec) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) spec_dst[bin_of<R3, LG>(lt, r)] = P[r];
                }
                if (to_tail) {
                    if (!FLAGS || __builtin_amdgcn_ballot_w64(tail_mask != 0xFFFFu) == 0) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) tail_dst[bin_of<R3, LG>(lt, r)] = P[r];
                    } else if (__builtin_amdgcn_ballot_w64(tail_mask != 0u) != 0) {
                        int lt_t = lt;  // (opaque: the sixteen addresses are rebuilt here, not kept across the loop)
                        asm volatile("" : "+v"(lt_t));
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            if ((tail_mask >> r) & 1u) tail_dst[bin_of<R3, LG>(lt_t, r)] = P[r];
                        }
                    }
                }
            } else {
                // Larger N: a lane's 16 bins are N/16 (or more) apart and neighbouring lanes' bins 64..512 B
                // apart, so direct stores would touch one cache line per lane.  The row goes through the
                // group's (now free) exchange rows instead and leaves as whole lines.  The barrier inside
                // must be reached by every wave of the workgroup: the decision is made for the workgroup
                // (its last group holds the latest segment), the stores stay per group.
                bool need;
                if constexpr (MODE == 1 || MODE == 2) {
                    need = true;
                } else if constexpr (LG > 64) {
                    const int seg_last = (cb * GPW + GPW - 1) * L + L - i;
                    need = !halo && seg_last >= T - p.tail_cols;
                    if constexpr (FLAGS) {
                        // whole rows, for the workgroup: some lane's tail_mask is not empty (word i mod 3 was set
                        // at the end of the step before, behind this step's barriers; the word read in the step
                        // before is cleared for the step after)
                        need = need && tail_any[i % 3] != 0u;
                        if (tid == 0) tail_any[(i + 2) % 3] = 0u;
                    }
                } else {
                    need = to_tail && (!FLAGS || __builtin_amdgcn_ballot_w64(tail_mask != 0u) != 0);  // whole rows, for the wave
                }
                if (need) {
                    float *row = reinterpret_cast<float *>(gx);  // N floats (plus the skew) of the group's LG * 36
                    // element i of the row sits at i + kSkew * (i / 32): the stores below go to bins 16 apart per lane
                    // (8-way conflicted at nperseg 4096 without the skew; tools/lds_banks.py rules), the loads to 32
                    // consecutive elements (conflict-free with any skew)
                    constexpr int kSkew = (R3 == 16) ? 2 : (R3 == 8) ? 4 : (R3 == 4) ? 1 : 0;
                    group_sync<LG>();  // every wave of the group is done with its exchange rows
                    int lt_w = lt;  // opaque, so that the sixteen lane-constant row offsets are not hoisted out of the step loop
                    asm volatile("" : "+v"(lt_w));
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int bin = bin_of<R3, LG>(lt_w, r);
                        row[bin + kSkew * (bin >> 5)] = P[r];
                    }
                    group_sync<LG>();
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const float q = row[j * LG + lt + kSkew * ((j * LG + lt) >> 5)];
                        if (to_spec) spec_dst[j * LG + lt] = q;
                        if (to_tail) tail_dst[j * LG + lt] = q;
                    }
                    wave_sync();  // (across waves: the next step's mean-reduction barrier)
                }
            }
        }

        RT_STAMP(10);  // detrend correction, power, row sums, tail columns
        if constexpr (RT_ABLATE == 7) continue;  // + power, row sums, tail columns (no candidate test)
        if constexpr (EMIT || FLAGS) {
            // candidates are rare: one max over the lane's 16 cells and a single compare in the
            // common path, the per-cell tests only where that fires.  (A NaN cell means the whole
            // segment is NaN -- the mean is -- so the max is NaN and `!(m < thr)` holds, as for the
            // reference's `not (P < thr)`.)
            // (MODE 6 with staged thresholds: the bits of the absolute threshold alone are built in the items that need them and in one
            // step of P elsewhere (rt_core.h: abs_sample_period -- eight, fewer where chunks are shorter) -- wave-uniform, so that the
            // other steps skip the block, with a phase that turns from item to item and wave to wave: a pulse train whose period is a
            // multiple of eight hops cannot hide from the count or fill it)
            const bool sampled_abs = THR_LDS && !item_hot && p.abs_hot && abs_sampled(i, samp_phase, samp_period) && !halo;  // (see item_hot)
            const bool want_hot = item_hot || sampled_abs;  // wave-uniform
            float mx = 0.f;
            if (want_hot) {
                mx = __builtin_fmaxf(__builtin_fmaxf(P[0], P[1]), P[2]);
#pragma unroll
                for (int r = 3; r < 15; r += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, P[r]), P[r + 1]);
                mx = __builtin_fmaxf(mx, P[15]);
            }
            uint32_t hot = 0;
            if (active && want_hot && !(mx < thr)) {
                // bit r = !(P[r] < thr), built by shifting (v_lshl_or_b32): `hot |= 1u << r` made hipcc keep the nine
                // literals 128 .. 32768 in VGPRs across the whole step loop (a select cannot take a literal on gfx9)
#pragma unroll
                for (int r = 15; r >= 0; --r) hot = (hot << 1) | ((P[r] < thr) ? 0u : 1u);
            }
            if constexpr (FLAGS) {
                // pre-filter bits: every cell of the chunk so far at or above the threshold; the run through t = 0
                // (it may continue a run of the previous buffer) counts whatever its length
                if constexpr (MODE == 6) {
                    uint32_t bits = 0;
                    int lt_b = lt;  // (lane index opaque: the addresses stay out of the loop's registers)
                    asm volatile("" : "+v"(lt_b));
                    if (active && !halo) {
                        bits = hot;
                        if (THR_LDS && p.thr_bin) {
                            // (the staged table holds max(absolute threshold, the bin's own): the one test)
                            // Every cell, every step, no look at the lane's maximum first: this level runs where the noise passes the
                            // absolute threshold in nearly every lane (on clean input the sparse level does the work), so the maximum
                            // only cost its fifteen instructions.
                            bits = 0;
#pragma unroll
                            for (int q = 3; q >= 0; --q) {
                                const float4 t4 = thr_lds[q * LG + lt_b];
                                bits = shift_in4(bits, P[4 * q + 0], P[4 * q + 1], P[4 * q + 2], P[4 * q + 3], t4.x, t4.y, t4.z, t4.w);
                            }
                        } else if (p.thr_bin && bits) {
                            // the bin's own second threshold (a lower bound of snr * row mean, see make_bin_thresholds):
                            // only cells that pass it too can be part of a plateau.  Sixteen floats per lane from L2.
                            const float4 *tb = reinterpret_cast<const float4 *>(p.thr_bin + ((int64_t)s * LG + lt_b) * 16);
                            uint32_t ok = 0;
#pragma unroll
                            for (int q = 3; q >= 0; --q) {
                                const float4 t4 = THR_LDS ? thr_lds[q * LG + lt_b] : tb[q];
                                ok = (ok << 1) | ((P[4 * q + 3] < t4.w) ? 0u : 1u);
                                ok = (ok << 1) | ((P[4 * q + 2] < t4.z) ? 0u : 1u);
                                ok = (ok << 1) | ((P[4 * q + 1] < t4.y) ? 0u : 1u);
                                ok = (ok << 1) | ((P[4 * q + 0] < t4.x) ? 0u : 1u);
                            }
                            bits &= ok;
                        }
                        if constexpr (!BITS_LDS && !(RT_EXP6 & 1)) p.cell_hot[((int64_t)s * T + seg) * LG + lt_b] = (uint16_t)bits;
                    }
                    if constexpr (BITS_LDS && !(RT_EXP6 & 1)) {
                        // steps 1 .. L (no halo step in this mode): slot (i - 1) & 3
                        // (the rows leave at the head of the step after their fourth, just ahead of that step's loads: flush_bits)
                        bits_lds[((i - 1) & 3) * LG + lt_b] = (uint16_t)bits;
                    }
                }
                if (active && !halo) {
                    if constexpr (MODE == 4 || MODE == 6) n_abs += (uint32_t)__builtin_popcount(hot) * ((THR_LDS && !item_hot) ? (uint32_t)samp_period : 1u);
                    allhot &= hot;
                    if (chunk == 0 && p.full) {  // (lane index opaque: the address stays out of the loop's registers)
                        int lt_f = lt;
                        asm volatile("" : "+v"(lt_f));
                        p.first[((int64_t)s * L + seg) * LG + lt_f] = (uint16_t)hot;
                    }
                }
                if constexpr (LG > 64 && SUMS) {
                    if (allhot != 0u && chunk_ok) tail_any[(i + 1) % 3] = 1u;  // the next step may have tail cells to write
                }
            }
            // a cell is kept if it is a candidate itself or directly precedes one (T11)
            const uint32_t emit = LISTED ? (active ? need_seg : 0u)  // the cells plan_runs asked for, whether they pass the threshold or precede one that does
                                         : (EMIT && active) ? ((halo ? (BELOW ? (next_hot & ~hot) : 0u) : (hot | next_hot)) & need_seg) : 0u;
            if (EMIT && RT_ABLATE != 9 && !gave_up && __builtin_amdgcn_ballot_w64(emit != 0) != 0) {  // wave-uniform, rare (RT_ABLATE 9: test without emission)
                // Candidates are staged per wave in LDS and flushed with ONE returned atomic per
                // flush: an atomic per cell would stall on vmcnt(0) and drain the prefetch.
                // (the lane index is made opaque here: otherwise hipcc hoists the sixteen lane-constant key bases
                // bin_of(lt, r) << tbits out of the step loop, where they cost registers the hot path is short of --
                // the mere presence of this rare block made the kernel 4 % slower at nperseg 256 and 8 % at 1024)
                int lt_e = lt;
                asm volatile("" : "+v"(lt_e));
                // Lane-centric: a lane counts its own cells, the counts' exclusive prefix over the wave and their total come from
                // the five bit planes of the counts (<= 16), and every lane stores its cells -- one or two where a tag sits -- in a
                // loop as long as the busiest lane's count, the power picked out of the registers by a select tree.  (Register by
                // register -- sixteen ballots for the total, sixteen more with a rank and a divergent store each -- the block was
                // ~260 instructions of every emitting step, and at config 2 every second wave step emits.)
                const int cnt = __builtin_popcount(emit);
                int off = 0, need = 0;  // this lane's first place; cells this wave emits in this step
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(((cnt >> k) & 1) != 0);
                    off += (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0)) << k;
                    need += __builtin_popcountll(m) << k;
                }
                if (stg_n + need > kStageLimit) {
                    flush_stage(p, s, stg, stg_n);
                    stg_n = 0;
                }
                if (need <= kStageLimit) {
                    uint32_t e = emit;
                    int o = stg_n + off;
                    while (__builtin_amdgcn_ballot_w64(e != 0u) != 0ull) {  // (wave-uniform)
                        if (e) {
                            const int r = __builtin_ctz(e);
                            e &= e - 1u;
                            const uint32_t key = ((uint32_t)bin_of<R3, LG>(lt_e, r) << p.tbits) | (uint32_t)seg;
                            stg[o++] = make_uint2(key, __float_as_uint(pick_range<0, 16>(P, r)));
                        }
                    }
                    stg_n += need;
                } else {
                    // more than a staging area in one step (dense input): straight to memory -- unless one of the
                    // stream's lists has overflowed already (count > capacity): then the call is re-run dense
                    // (AUTO) or fails (SPARSE) whatever else is emitted, and an atomic per cell on the 16 counters
                    // of a stream (2 M cells, ~10 ms per batch of all-hot input) would only delay that
                    const int lane_ = threadIdx.x & 63;
                    const uint32_t cnt = lane_ < kBuckets ? __hip_atomic_load(&p.hot_count[s * kBuckets + lane_], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                    gave_up = __builtin_amdgcn_ballot_w64(cnt > (uint32_t)p.hot_cap) != 0;
#pragma unroll
                    for (int r = 0; r < 16 && !gave_up; ++r) {
                        if (emit & (1u << r)) {
                            const int bin = bin_of<R3, LG>(lt_e, r);
                            const int bkt = bin & (kBuckets - 1);
                            const uint32_t slot = atomicAdd(&p.hot_count[s * kBuckets + bkt], 1u);
                            if (slot < (uint32_t)p.hot_cap) {
                                const uint32_t key = ((uint32_t)bin << p.tbits) | (uint32_t)seg;
                                p.hot[((int64_t)s * kBuckets + bkt) * p.hot_cap + slot] = make_uint2(key, __float_as_uint(P[r]));
                            }
                        }
                    }
                }
            }
            next_hot = hot;
            if constexpr (EMIT && BELOW && !LISTED) {
                if (i == L) {
                    const bool below = chunk_ok && c0 > 0 && (hot & need) != 0u;  // a lowest cell of the chunk is a candidate
                    bool any_below;
                    if constexpr (LG > 64) {
                        any_below = __syncthreads_or(below) != 0;  // the step has workgroup barriers
                    } else {
                        any_below = __builtin_amdgcn_ballot_w64(below) != 0;
                    }
                    if (any_below) {
                        n_steps = L + 1;
                        request_segment(c0 - 1);
                    }
                }
            }
        }
    }

#ifdef RT_STAMPS
    RT_STAMP(0);
    st_acc[11] = (uint32_t)(n_steps - i_first + 1);
    st_acc[12] = st_prev - st_t0;
    st_acc[13] = (uint32_t)__builtin_amdgcn_s_memrealtime() - st_r0;
    st_acc[14] = st_wave_start;
    st_acc[15] = (uint32_t)__builtin_amdgcn_s_memrealtime();
    if (p.dbg && (threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < kStamps; ++k) p.dbg[((int64_t)item * (BLK / 64) + (threadIdx.x >> 6)) * kStamps + k] = st_acc[k];
    }
#endif
    if constexpr (EMIT) {
        if (stg_n) flush_stage(p, s, stg, stg_n);
    }
    if constexpr (BITS_LDS) flush_bits(n_steps, ((n_steps - 1) & 3) + 1);  // (the chunk's last one to four steps)
    if constexpr (FLAGS) {
        if (p.full && chunk_ok) p.full[((int64_t)s * p.chunks + chunk) * LG + lt] = (uint16_t)(allhot & 0xFFFFu);
    }
    if constexpr (MODE == 4 || MODE == 6) {
        if (p.abs_hot && !(RT_EXP6 & 8)) {  // (once per item)
            uint32_t n = n_abs;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) n += (uint32_t)__shfl_xor((int)n, o);
            if ((threadIdx.x & 63) == 0 && n) atomicAdd(&p.abs_hot[s], n);
        }
    }
    if constexpr (MODE == 3) {
        if (acc[0] == 12345.678f) p.psum[0] = acc[0];  // keeps the loads alive, never true in practice
    } else if constexpr (SUMS) {
        if constexpr (LIN) {
            if (p.dc_flag) {
                // the quietest bin of this lane (bins 0 and +-1 carry what the detrend left) and the lane's total, then the wave's
                // (wave-wide figures: at nperseg <= 512 a wave holds several chunks of the stream, from nperseg 2048 on a part of the bins)
                float mn = 3.0e38f, tot = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool dc_bin = (r == kS0.reg && lt == kS0.lane) || (r == kS1.reg && lt == kS1.lane) || (r == kS2.reg && lt == kS2.lane);
                    mn = fminf(mn, (dc_bin || !chunk_ok) ? 3.0e38f : acc[r]);
                    tot += chunk_ok ? acc[r] : 0.f;
                }
                float dsum = chunk_ok ? dc_acc : 0.f;  // (the same in the LG lanes of a group)
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    mn = fminf(mn, __shfl_xor(mn, o));
                    tot += __shfl_xor(tot, o);
                    dsum += __shfl_xor(dsum, o);
                }
                constexpr int kLanesPerGroup = LG < 64 ? LG : 64;
                const float groups = (float)__builtin_popcountll(__builtin_amdgcn_ballot_w64(chunk_ok)) / (float)kLanesPerGroup;
                const float d = dsum / (float)kLanesPerGroup;  // D summed over the wave's groups
                if (d > p.dc_limit * mn * groups && d > p.dc_limit2 * tot && (threadIdx.x & 63) == 0) p.dc_flag[s] = 1;
            }
        }
        // deterministic workgroup reduction of the lane groups' row sums: one
        // partial row per workgroup (fixed summation order, no float atomics)
        __syncthreads();
        float *part = reinterpret_cast<float *>(xch);  // [GPW][N] floats = 16 KiB
#pragma unroll
        for (int r = 0; r < 16; ++r) part[g * N + bin_of<R3, LG>(lt, r)] = chunk_ok ? acc[r] : 0.f;
        __syncthreads();
        float *dst = p.psum + ((int64_t)s * p.blocks_per_stream + cb) * N;
        if constexpr (N < BLK) {
            // nperseg < 256 (QS): BLK / N threads per bin, each over its sixteen lane groups in order; their sums (and minima) through
            // LDS behind the rows just read, added in order by the first of them -- a fixed order again
            constexpr int PARTS = BLK / N, PERP = GPW / PARTS;
            static_assert(PERP == 16, "sixteen lane groups per part at every QS");
            float *const part2 = part + GPW * N;  // [PARTS][N] sums, then [PARTS][N] minima (the exchange block holds >= 2 GPW N floats)
            const int bin = tid % N, hpart = tid / N;
            float sum = 0.f;
#pragma unroll
            for (int gg = 0; gg < PERP; ++gg) sum += part[(hpart * PERP + gg) * N + bin];
            part2[hpart * N + bin] = sum;
            if (p.chunk_min && !(RT_EXP6 & 2)) {
                const int grp = minsum_group(L, PERP);  // (a power of two <= 16: the parts hold whole groups)
                float mn = 3.0e38f, run = 0.f;
                bool whole = true;
#pragma unroll
                for (int gg = 0; gg < PERP; ++gg) {
                    const int ch = cb * GPW + hpart * PERP + gg;
                    whole = whole && (ch < p.chunks && (ch + 1) * L <= T);
                    run += part[(hpart * PERP + gg) * N + bin];
                    if (((gg + 1) & (grp - 1)) == 0) {
                        if (whole) mn = fminf(mn, run);
                        run = 0.f;
                        whole = true;
                    }
                }
                part2[(PARTS + hpart) * N + bin] = mn;
            }
            __syncthreads();
            if (tid < N) {
                float tot = part2[tid];
#pragma unroll
                for (int h2 = 1; h2 < PARTS; ++h2) tot += part2[h2 * N + tid];
                dst[tid] = tot;
                if (p.chunk_min && !(RT_EXP6 & 2)) {
                    float mn = part2[PARTS * N + tid];
#pragma unroll
                    for (int h2 = 1; h2 < PARTS; ++h2) mn = fminf(mn, part2[(PARTS + h2) * N + tid]);
                    p.chunk_min[((int64_t)s * p.blocks_per_stream + cb) * N + tid] = (mn < 3.0e38f) ? __float_as_uint(mn) : 0x7f7f7f7fu;
                }
            }
        } else {
#pragma unroll
        for (int j = 0; j < N / BLK; ++j) {
            const int bin = tid + BLK * j;
            float sum = 0.f;
#pragma unroll
            for (int gg = 0; gg < GPW; ++gg) sum += part[gg * N + bin];
            dst[bin] = sum;
            if (p.chunk_min && !(RT_EXP6 & 2)) {
                // the quietest complete run of >= 32 segments of the bin in this item: a chunk, or a group of consecutive chunks
                // where chunks are shorter (rt_core.h: minsum_group)
                const int grp = minsum_group(L, GPW);
                float mn = 3.0e38f, run = 0.f;
                bool whole = true;
#pragma unroll
                for (int gg = 0; gg < GPW; ++gg) {
                    const int ch = cb * GPW + gg;
                    whole = whole && (ch < p.chunks && (ch + 1) * L <= T);
                    run += part[gg * N + bin];
                    if (((gg + 1) & (grp - 1)) == 0) {
                        if (whole) mn = fminf(mn, run);
                        run = 0.f;
                        whole = true;
                    }
                }
                p.chunk_min[((int64_t)s * p.blocks_per_stream + cb) * N + bin] = (mn < 3.0e38f) ? __float_as_uint(mn) : 0x7f7f7f7fu;
            }
        }
        }
    }
    if constexpr (!PERSIST) break;
    // next item: the ticket drawn at the start of this one
    __syncthreads();  // every wave is done with this item's LDS (exchange rows, row-sum scratch, item word)
    if (tid == 0) *item_word = ticket;
    __syncthreads();
    item = (int)gridDim.x + (int)*item_word;
  }
    if constexpr (PERSIST) {
        // the workgroup that leaves last puts the counters back for the next launch on this stream
        if (tid == 0) {
            const uint32_t left = atomicAdd(p.work + 1, 1u);
            if (left + 1u == gridDim.x) {
                atomicExch(p.work, 0u);
                atomicExch(p.work + 1, 0u);
            }
        }
    }
}

// Run-length pre-filter, between pass A and pass B, one workgroup per stream:
//   * the threshold bits of chunk 0 become "every cell from t = 0 up to this one passes" -- the cells of the runs
//     that may continue a run of the previous buffer, which count whatever their length (idempotent);
//   * the chunks pass B has to transform again (a set bit in the chunk itself or a neighbour, chunk 0 also for a hot
//     cell at t = 0) are packed `gpw` to a workgroup: in the noise-floor regime a fifth of the chunks is needed, spread
//     so that most workgroups of a one-to-one launch would keep one wave busy and three idle.
// Dynamic LDS: one bit per chunk.
__global__ __launch_bounds__(256) void plan_pass_b(const uint16_t *full, uint16_t *first, int32_t *item_chunks, int32_t *item_count,
                                                   int lg, int segs_per_chunk, int n_seg, int chunks, int blocks_per_stream, int gpw) {
    extern __shared__ uint32_t plan_lds[];
    uint32_t *const any_w = plan_lds;  // [(chunks + 31) / 32]
    __shared__ uint32_t first_any, wave_cnt[4];
    const int s = blockIdx.x, tid = threadIdx.x;
    const int words = (chunks + 31) / 32;
    for (int i = tid; i < words; i += 256) any_w[i] = 0u;
    if (tid == 0) first_any = 0u;
    __syncthreads();
    if (tid < lg) {
        uint16_t *f = first + (int64_t)s * segs_per_chunk * lg + tid;
        if (f[0]) atomicOr(&first_any, 1u);
        uint16_t m = 0xFFFFu;
        const int n = segs_per_chunk < n_seg ? segs_per_chunk : n_seg;
        for (int t = 0; t < n; ++t) {
            m &= f[(int64_t)t * lg];
            f[(int64_t)t * lg] = m;
        }
    }
    const uint16_t *fs = full + (int64_t)s * chunks * lg;
    for (int64_t i = tid; i < (int64_t)chunks * lg; i += 256) {
        if (fs[i]) {
            const int c = (int)(i / lg);
            atomicOr(&any_w[c >> 5], 1u << (c & 31));
        }
    }
    __syncthreads();
    auto bit = [&](int c) -> bool { return c >= 0 && c < chunks && ((any_w[c >> 5] >> (c & 31)) & 1u); };
    int32_t *items = item_chunks + (int64_t)s * blocks_per_stream * gpw;
    int base = 0;  // workgroup-uniform: needed chunks below this tile
    for (int c0 = 0; c0 < chunks; c0 += 256) {
        const int c = c0 + tid;
        const bool needed = c < chunks && (bit(c - 1) || bit(c) || bit(c + 1) || (c == 0 && first_any));
        const unsigned long long m = __builtin_amdgcn_ballot_w64(needed);
        if ((tid & 63) == 0) wave_cnt[tid >> 6] = (uint32_t)__builtin_popcountll(m);
        __syncthreads();
        int off = base;
        for (int w = 0; w < (tid >> 6); ++w) off += (int)wave_cnt[w];
        if (needed) items[off + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0))] = c;
        base += (int)(wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3]);
        __syncthreads();
    }
    const int n_items = (base + gpw - 1) / gpw;
    for (int i = base + tid; i < n_items * gpw; i += 256) items[i] = chunks;  // the last workgroup's idle lane groups
    if (tid == 0) item_count[s] = n_items;
}

// Exact run-length pre-filter (RT_MODE_RUNFILTER), between its two scans.  Where the chunk bits above need
// `signal_min_duration >= 2 L - 1` hops (the reference's default geometry, 300 kS/s: 8 ms = 9.4 hops, does not qualify for any
// useful L), this one works on the threshold bit of EVERY cell (MODE 6 writes them: 2 B per lane and segment, 1.6 % of the IQ
// bytes) and is exact for any chunk length.  A plateau can only become a signal if it has at least r = min_run_cells cells
// (the duration gate, rt_core.h: gate_run) or contains t = 0 (it may continue a plateau of the previous buffer), and every
// plateau lies inside a run of cells that pass the absolute threshold.  So the cells the detection needs are
//     C = { cells of threshold runs of length >= r }  u  { cells of the run through t = 0 },   plus the cell before each
// (`data` starts on it, analyze.py:382-398), and only the segments holding such cells are transformed again (MODE 7).
// Per bin these are window operations along t, done for the 16 bins of a lane at once on its 16-bit words:
//     E[t] = AND_{j<r} H[t+j]   (a window of r set bits starts at t)      by doubling: P_1 = H, P_2k[t] = P_k[t] & P_k[t+k]
//     C[t] = OR_{j<r}  E[t-j]   (t lies in such a window)                 likewise with OR and negative offsets
// Writes need[t] = C[t] | C[t+1] for the tile's rows and appends the rows with any bit set to the stream's segment list
// (order irrelevant: the detection sorts its cells).
// The 16-bit words of four neighbouring lanes are handled as one 64-bit word (the operations are bitwise, lg is a multiple of
// 16): a row is w = lg / 4 words.
//
// Streaming form, no LDS passes: a lane owns one word column of one tile of rows and walks it upwards in time with two
// bit-sliced counters (K bit planes of 64 independent counters each):
//     run[u]   = length of the threshold run ending at row u, as a sticky flag   SAT[u] = (run[u] >= r)
//     since[u] = rows since the last SAT row, as a flag                          FAR[u] = (since[u] >= r)
// SAT[u] says a window of r set bits ends at u, so row t lies in a run of >= r cells iff some u in t .. t + r - 1 has SAT,
// i.e. C[t] = ~FAR[t + r - 1]: every decision r - 1 rows behind the row just read, nothing kept but the counters.  Rows
// before the buffer count as set (a run through t = 0 may continue a plateau of the previous buffer: any length keeps it),
// rows past it as clear.  A tile reads r - 1 rows below and r above its own (the counters start empty / far).
// Before: the tile and its halo in LDS, van Herk / Gil-Werman window AND and OR in fifteen LDS passes with a wave barrier
// between them, one wave per workgroup -- 0.36 ms for the 157 MB of bits of 4 096 streams at the reference's default
// geometry, five times what its bytes take (profiles/r04_d_*); the first version (doubling, 16-bit words, tiles of 1 000
// rows in 100 KiB of LDS) took 3.3 ms, twice the scan it serves.
// One wave per workgroup: 64 / w tiles of `tile_rows` rows side by side (a lane per tile and word column; the wave's rows
// are contiguous), LDS only for a byte per row (does the row keep anything?), from which the stream's segment list is
// appended with one returned atomic per wave (order irrelevant: the detection sorts its cells).
// (kPlanRowsPerWave, kPlanMaxRun, RunPlanner, plan_tile_rows: rt_core.h -- shared with the host check of the CPU test-suite)
template <int K>
__global__ __launch_bounds__(64) void plan_runs(const uint16_t *hot, uint16_t *need, int32_t *seg_list, int32_t *seg_count,
                                                int n_seg, int lg, int r, int tile_rows) {
    extern __shared__ unsigned char plan_any[];  // [64 / w * tile_rows]
    using u64 = unsigned long long;
    const int s = blockIdx.x, lane = threadIdx.x;  // (streams along x: a grid's y extent ends at 65 535)
    const int w = lg / 4;                     // 64-bit words per row (4 .. 64)
    const int tpw = 64 / w;                   // tiles per wave
    const int c = lane % w, tj = lane / w;
    const int B = tile_rows;
    const int wave_rows = tpw * B;
    const int row0 = blockIdx.y * wave_rows;  // the wave's first row
    const int a = row0 + tj * B;              // this lane's tile: rows a .. a + B - 1
    for (int i = lane * 4; i < wave_rows; i += 256) *reinterpret_cast<uint32_t *>(plan_any + i) = 0u;  // (the block is a multiple of 4 bytes)
    wave_sync();
    const u64 *H = reinterpret_cast<const u64 *>(hot + (int64_t)s * n_seg * lg) + c;
    u64 *Nd = reinterpret_cast<u64 *>(need + (int64_t)s * n_seg * lg) + c;
    RunPlanner<K> pl;  // (rt_core.h: the two bit-sliced counters)
    pl.init(r);
    u64 c_prev = 0ull;
    const int t_end = (a + B < n_seg) ? a + B : n_seg;  // rows of this tile inside the buffer
    const int n_steps = B + 2 * r;  // rows a - r + 1 .. a + B + r - 1 (+ 1: a multiple of nothing in particular; the batches below round up)
    // rows in batches of eight, the next batch requested before the current one is counted
    auto load_rows = [&](int i0, u64 (&dst)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int u = a - r + 1 + i0 + j;
            dst[j] = (u < 0) ? ~0ull : (u < n_seg) ? H[(int64_t)u * w] : 0ull;
        }
    };
    u64 h[8], hn[8];
    load_rows(0, hn);
    for (int i0 = 0; i0 < n_steps; i0 += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) h[j] = hn[j];
        if (i0 + 8 < n_steps) load_rows(i0 + 8, hn);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int u = a - r + 1 + i0 + j;
            const u64 hh = h[j];
            const u64 c_now = pl.step(hh);  // C[u - r + 1]
            // need[t] = C[t] | C[t + 1] for t = u - r (the cell before a run: `data` starts on it)
            const int t = u - r;
            if (t >= a && t < t_end) {
                const u64 v = c_prev | ((t + 1 < n_seg) ? c_now : 0ull);
                // Only the words that keep anything are written (round 6): the array is all zeros between calls -- zeroed at creation,
                // and the listed scan puts every word it reads back to zero --, and of the 157 MB the planner wrote per 4 096 streams at
                // the reference's defaults 99 % were zeros.  (Were a word ever left over, its cells would be emitted IN ADDITION: the
                // detection evaluates every cell it is given, and a superset of the kept cells gives the same records.)
                if (v) {
                    Nd[(int64_t)t * w] = v;
                    plan_any[t - row0] = 1;  // (benign race: every writer stores 1)
                }
            }
            c_prev = c_now;
        }
    }
    wave_sync();
    // the wave's rows that keep anything go to the stream's list: counted, places reserved by ONE returned atomic, then written
    int total = 0;
    for (int b0 = 0; b0 < wave_rows; b0 += 64) {  // (wave-uniform bounds)
        const bool has = b0 + lane < wave_rows && plan_any[b0 + lane];
        total += __builtin_popcountll(__builtin_amdgcn_ballot_w64(has));
    }
    if (total == 0) return;  // (wave-uniform)
    int base = 0;
    if (lane == 0) {
        base = atomicAdd(&seg_count[s], total);
        // the batch's total (word [S] of the counts) tells the host how selective the level is on this input
        atomicAdd(&seg_count[gridDim.x], total);
    }
    base = __builtin_amdgcn_readfirstlane(base);
    for (int b0 = 0; b0 < wave_rows; b0 += 64) {
        const bool has = b0 + lane < wave_rows && plan_any[b0 + lane];
        const unsigned long long m = __builtin_amdgcn_ballot_w64(has);
        if (has) seg_list[(int64_t)s * n_seg + base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0))] = row0 + b0 + lane;
        base += __builtin_popcountll(m);
    }
}

// The largest per-stream count of cells at or above the absolute threshold (StftParams::abs_hot, left by a MODE 4 / 6
// scan) -> pinned host word; the counts are put back to zero for the next call.  One workgroup.
__global__ __launch_bounds__(256) void max_abs_hot(uint32_t *abs_hot, int n_streams, uint32_t *host_max) {
    __shared__ uint32_t wave_max[4];
    uint32_t m = 0;
    for (int s = threadIdx.x; s < n_streams; s += 256) {
        m = max(m, abs_hot[s]);
        abs_hot[s] = 0u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) *host_max = max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3]));
}

// first segment and length of chunk `c` of a stream whose earliest n_short chunks are short (StftParams::short_chunks)
struct ChunkSpan {
    int c0, len;
};
__host__ __device__ inline ChunkSpan chunk_span(int c, int L, int n_short, int short_len) {
    if (c < n_short) return ChunkSpan{c * short_len, short_len};
    return ChunkSpan{n_short * short_len + (c - n_short) * L, L};
}
// The geometry for n_seg segments: a quarter of the stream (its earliest part) in chunks of half the length.  With items drawn
// latest chunks first from one counter, the waves of a launch finish within one item's length of one another; half-length items
// at the end halve that tail (stft_scan64, config-5 share: 1.85 of 2 wave slots busy over the launch with equal items).
__host__ __device__ inline void chunk_geometry(int n_seg, int L, bool two_level, int *n_short, int *short_len, int *chunks) {
    const int long0 = (n_seg + L - 1) / L;
    if (!two_level || long0 < 4 || L < 8) {
        *n_short = 0;
        *short_len = L;
        *chunks = long0 > 0 ? long0 : 0;
        return;
    }
    const int k = long0 / 4;            // long chunks' worth of segments given to short ones
    *short_len = (L + 1) / 2;
    *n_short = 2 * k;
    const int rest = n_seg - *n_short * *short_len;
    *chunks = *n_short + (rest > 0 ? (rest + L - 1) / L : 0);
}

// Which kernel serves nperseg 4096: stft_scan64 (rt_scan64.h: one wave per segment, 64 bins per lane) -- the default -- or,
// with -DRT_WAVE64_4096=0, the four-waves-per-segment instantiation stft_scan<16, ...> it replaced (kept for A/B runs).
#ifndef RT_WAVE64_4096
#define RT_WAVE64_4096 1
#endif
__host__ __device__ constexpr bool scan_wave64(int R3) { return RT_WAVE64_4096 && R3 == 16; }

// bin of result register r in lane lt of a group, R3 at run time (bin_of<R3>)
__device__ __forceinline__ int bin_of_rt(int R3, int lg, int lt, int r) {
    if (R3 == 1) return lt + lg * r;
    const int G = 16 / R3;
    const int k1 = lt / R3, qg = lt % R3;
    const int u = (r / R3 - ((k1 * R3 / 8) & (G - 1))) & (G - 1), q2 = r % R3;
    return k1 + 16 * (qg * G + u) + 256 * q2;
}

// the bin whose threshold sits at position j of a stream's table in lane order (StftParams::thr_bin): [lane][16 registers] of the
// 16-points-per-lane scans, [lane][64 registers] of stft_scan64 (bin = lane + 64 register)
__device__ __forceinline__ int lane_order_bin(int R3, int lg, int j) {
    if (scan_wave64(R3)) return j / 64 + 64 * (j % 64);
    return bin_of_rt(R3, lg, j / 16, j % 16);
}

// Per-bin thresholds for the exact pre-filter's bits (MODE 6), from the PREVIOUS call's chunk minima.  The reference's
// predicate is `!(P < thr) && !(P / row_mean < snr)` (analyze.py:370, 378); with the noise floor over the absolute
// threshold the first test says nothing and the second decides -- but the row mean of a buffer is known only after its
// scan.  The quietest complete chunk of the previous buffer (where chunks are shorter than 32 segments: the quietest group
// of consecutive chunks that makes up 32, rt_core.h: minsum_group; `L` below is that length) gives a lower bound that
// survives tags coming and going (a pulse sits in one or two chunks of dozens): theta = snr * (smallest sum / L) is about
// 0.55 * snr * (noise mean) for stationary noise.  A cell passes the full predicate only if it passes `P >= theta` -- PROVIDED theta <= snr * (this
// buffer's row mean), which check_bin_thresholds verifies after the scan; a stream that fails it (its floor dropped by
// more than ~2.5 dB from one buffer to the next) is analysed again (rt_fetch: a few streams dense, else the call on its own
// row means).  No estimate (first call, buffers shorter than a chunk): theta = 0, the bits are the absolute threshold's alone.
__global__ __launch_bounds__(256) void make_bin_thresholds(const uint32_t *chunk_min_prev /* [S][prev_items][N] */, int prev_items, float *thr_bin /* lane order */,
                                                          float *thr_nat /* [S][N] */, int n_streams, int R3, int lg, int L, float snr) {
    const int N = 16 * lg;  // (every scan holds sixteen bins per lane -- stft_scan64: 64 per lane, lg = 256 there as well)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over [S][LG][16]
    if (i >= (int64_t)n_streams * N) return;
    const int s = (int)(i / N);
    const int bin = lane_order_bin(R3, lg, (int)(i % N));
    float th = 0.f;
    if (chunk_min_prev) {
        uint32_t m = 0x7f7f7f7fu;  // (positive floats order like their bits)
        for (int c = 0; c < prev_items; ++c) m = min(m, chunk_min_prev[((int64_t)s * prev_items + c) * N + bin]);
        const float mn = __uint_as_float(m);
        if (mn < 1.0e38f && mn == mn) th = snr * minsum_margin(L) * (mn / (float)L);
    }
    thr_bin[i] = th;
    thr_nat[(int64_t)s * N + bin] = th;
}

// The same table from THIS buffer's row means (a call analysed again after its thresholds failed the check below: the
// failed scan left the partial row sums): theta = snr * row_mean * (1 - 1e-6), the bound itself -- the check cannot fail.
__global__ __launch_bounds__(256) void make_bin_thresholds_from_means(const float *psum, int items_per_stream, int n_seg, float *thr_bin, float *thr_nat,
                                                                     int n_streams, int R3, int lg, float snr) {
    const int N = 16 * lg;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over [S][LG][16]
    if (i >= (int64_t)n_streams * N) return;
    const int s = (int)(i / N);
    const int bin = lane_order_bin(R3, lg, (int)(i % N));
    double sum = 0.0;
    for (int c = 0; c < items_per_stream; ++c) sum += (double)psum[((int64_t)s * items_per_stream + c) * N + bin];
    const float avg = (float)sum / (float)n_seg;
    float th = snr * avg * (1.0f - 1.0e-6f);
    if (!(th > 0.f) || !(th < 3.0e38f)) th = 0.f;  // (NaN / overflowing sums: the absolute threshold alone)
    thr_bin[i] = th;
    thr_nat[(int64_t)s * N + bin] = th;
}

// ... and the check behind the scan: theta <= snr * row_mean * (1 - 1e-6) for every bin of every stream (the row mean as
// the detection computes it, from the same partial sums).  A stream that fails is marked like one whose candidate lists
// overflowed: rt_fetch re-runs it dense (a few streams) or takes the batch one level up.
__global__ __launch_bounds__(256) void check_bin_thresholds(const float *thr_nat, const float *psum, int n_streams, int N, int items_per_stream, int n_seg,
                                                           float snr, int32_t *stream_overflow, unsigned long long *counters, unsigned long long flag) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)n_streams * N) return;
    const float th = thr_nat[i];
    if (!(th > 0.f)) return;
    const int s = (int)(i / N), bin = (int)(i % N);
    double sum = 0.0;
    for (int c = 0; c < items_per_stream; ++c) sum += (double)psum[((int64_t)s * items_per_stream + c) * N + bin];
    const float avg = (float)sum / (float)n_seg;
    if (!(th <= snr * avg * (1.0f - 1.0e-6f))) {
        stream_overflow[s] = 1;
        atomicOr(counters + 2, flag);
    }
}

// What stands between the exact pre-filter's first scan and its planning kernel, in ONE launch (round 5; before: max_abs_hot,
// check_bin_thresholds and a memset, three launches of 5 - 8 us with their gaps on a stream that has nothing else to do):
// the check above for every (stream, bin); the segment counters of the planner back to zero (words 0 .. n_streams of
// `seg_count`); and, by workgroup 0, max_abs_hot's job.
__global__ __launch_bounds__(256) void after_bit_scan(uint32_t *abs_hot, uint32_t *host_max, const float *thr_nat, const float *psum, int n_streams, int N,
                                                     int items_per_stream, int n_seg, float snr, int32_t *stream_overflow, unsigned long long *counters,
                                                     unsigned long long flag, int32_t *seg_count) {
    __shared__ uint32_t wave_max[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i <= n_streams) seg_count[i] = 0;
    if (blockIdx.x == 0) {
        uint32_t m = 0;
        for (int s = threadIdx.x; s < n_streams; s += 256) {
            m = max(m, abs_hot[s]);
            abs_hot[s] = 0u;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
        if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) *host_max = max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3]));
    }
    if (i >= (int64_t)n_streams * N) return;
    const float th = thr_nat[i];
    if (!(th > 0.f)) return;
    const int s = (int)(i / N), bin = (int)(i % N);
    double sum = 0.0;
    for (int c = 0; c < items_per_stream; ++c) sum += (double)psum[((int64_t)s * items_per_stream + c) * N + bin];
    const float avg = (float)sum / (float)n_seg;
    if (!(th <= snr * avg * (1.0f - 1.0e-6f))) {
        stream_overflow[s] = 1;
        atomicOr(counters + 2, flag);
    }
}

// ---------------------------------------------------------------------------
// detection
// ---------------------------------------------------------------------------
constexpr int kDetBlock = 1024;  // 16 waves: one stream per workgroup

struct DetectArgs {
    DetectParams dp;
    int32_t n_streams;
    int32_t n_bins;            // F
    // previous-buffer cells: prev[(s*prev_cols + (prev_cols - d))*F + f], d >= 1
    const float *prev;
    int32_t prev_cols;
    // sparse inputs
    const uint2 *hot;          // [S][kBuckets][hot_cap]
    const uint32_t *hot_count; // [S][kBuckets]
    uint32_t *hot_count_rw;    // same array, zeroed by its last reader
    uint32_t *large_any;       // [S] a bucket of the stream has more than kSmallBucket cells (small instantiation -> large one; zeroed by finalize_records)
    int32_t *hot_total;        // [S] (host-visible) candidate cells per stream
    int32_t lds_cells;         // cells the large instantiation's LDS holds (power of two)
    int32_t cand_cap;          // plateaus a wave can stage per bucket (LDS)
    int32_t hot_cap;           // cells per (stream, bucket)
    int32_t tbits;             // key = bin << tbits | t
    rt_record *raw;            // [S][rec_cap] unordered records of the bucket waves
    int32_t *raw_count;        // [S]
    const float *psum;         // [S][chunks][F] partial row sums
    int32_t chunks;            // partial rows per stream
    // dense input
    const float *spec;         // [S][T][F]
    // outputs
    rt_record *records;        // pool
    int64_t pool_cap;
    int32_t rec_cap;           // per stream
    int32_t *rec_offset;       // [S]
    int32_t *rec_count;        // [S]
    unsigned long long *counters;  // [0] records allocated, [1] hot total, [2] flags, [3] workgroups done (close_call), [4] the most records any stream wanted
    unsigned long long *host_counters;  // pinned host copy of the kCounterWords words, written by the call's last workgroup
    // per-stream overrides (null = dp's value for every stream)
    const float *thr_s;        // [S] signal_threshold of the stream's SDR (analyze.py:115)
    const float *cal_s;        // [S] its calibration_db (orders maxima in the shadow filter)
    const int32_t *no_last;    // [S] (host-visible) non-zero: this stream has no previous buffer in this call
                               //     (a restarted SDR's fresh analyzer, analyze.py:128)
    const int32_t *stream_list;  // detect_dense: null, or the streams of this launch (spectrogram indexed by position, see StftParams)
    int32_t *stream_overflow;  // [S] (host-visible) set for a stream one of whose candidate lists overflowed
    int32_t *stream_incons;    // [S] (host-visible) set for a stream in which a run lacked its preceding cell: an internal error,
                               //     unless the stream overflowed (the scan stops emitting for such a stream, its other lists are torn)
    const int32_t *seg_total;  // null, or the device word holding the number of segments the exact pre-filter's planner listed over the batch ...
    int32_t *host_seg_total;   // ... and the pinned host word the call's last finalize_records workgroup copies it to (how selective the level was: AUTO)
    int32_t *work_list;        // null, or (detect_group) [S * kQuarters] the (stream, quarter) pairs left to the per-bucket waves ...
    int32_t *work_count;       // ... and their number (zeroed by finalize_records)
    int32_t filtered;          // the candidate lists come from the run-length pre-filter (stft_scan MODE 5): a run whose
                               //     preceding cell is missing lies across the edge of the emitted chunks, is too short
                               //     to pass the duration gate and is dropped (without the filter that is an internal error)
};

// the detect parameters as stream `s` sees them (s is workgroup- or wave-uniform: scalar loads)
__device__ __forceinline__ DetectParams stream_params(const DetectArgs &a, int s) {
    DetectParams dp = a.dp;
    if (a.thr_s) dp.thr = a.thr_s[s];
    if (a.cal_s) dp.cal_db = a.cal_s[s];
    if (a.no_last && a.no_last[s]) dp.n_seg_last = -1;
    return dp;
}

constexpr int kCounterWords = 5;
// Records a stream may have on the dense path while its unordered list is staged in LDS (detect_dense<false>: 56 bytes each);
// a handle whose record capacity has grown beyond it -- the reference appends without limit, analyze.py:449-450 -- stages in
// global memory instead and leaves ranking and shadow verdicts to finalize_records (detect_dense<true>).
constexpr int kDenseLdsRecords = 2048;
constexpr unsigned long long kFlagHotOverflow = 1ull;
constexpr unsigned long long kFlagRecOverflow = 2ull;
constexpr unsigned long long kFlagInconsistent = 4ull;
constexpr unsigned long long kFlagThrStale = 8ull;  // check_bin_thresholds: a stream's per-bin thresholds were too high for this buffer (comes with kFlagHotOverflow)

// sum of one bin's partial row sums in float64, in partial-row order; loads are issued
// eight at a time so their latency overlaps (the additions keep the sequential order)
__device__ __forceinline__ double row_sum_from_partials(const float *ps, int chunks, int F) {
    double sum = 0.0;
    int c = 0;
    for (; c + 8 <= chunks; c += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ps[(int64_t)(c + j) * F];
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += (double)v[j];
    }
    for (; c < chunks; ++c) sum += (double)ps[(int64_t)c * F];
    return sum;
}

struct PrevCells {
    const float *base;  // points at column prev_cols of this (stream, bin): base[-d*F]
    int32_t F;
    __device__ float operator()(int32_t d) const { return base[-(int64_t)d * F]; }
};

// LDS record staging shared by both detect kernels
struct RecLds {
    rt_record *rec;       // [rec_cap]
    long long *ts_us;     // [rec_cap]
    long long *dur_us;    // [rec_cap]
    int *count;           // [1] (+ scratch words)
};

__device__ __forceinline__ RecLds carve_rec_lds(unsigned char *&ptr, int rec_cap) {
    RecLds l;
    l.ts_us = reinterpret_cast<long long *>(ptr);
    ptr += sizeof(long long) * rec_cap;
    l.dur_us = reinterpret_cast<long long *>(ptr);
    ptr += sizeof(long long) * rec_cap;
    l.rec = reinterpret_cast<rt_record *>(ptr);
    ptr += sizeof(rt_record) * rec_cap;
    l.count = reinterpret_cast<int *>(ptr);
    ptr += 16;
    return l;
}

// phase 1: a gated run becomes a record without statistics; `cell_off` tells
// phase 2 where the run's cells are (sparse: index offset into the sorted list)
__device__ __forceinline__ void push_candidate(const DetectArgs &a, RecLds &l, int s, int fi, int start, int end,
                                               float avg, int cell_off) {
    const int idx = atomicAdd(l.count, 1);
    if (idx < a.rec_cap) {
        rt_record r;
        r.stream = s;
        r.fi = fi;
        r.start = start;
        r.end = end;
        r.max_p = 0.f;
        r.mean_p = 0.f;
        r.std_db = 0.f;
        r.row_mean = avg;
        r.shadowed = 0;
        r.reserved = cell_off;
        l.rec[idx] = r;
        if (l.ts_us) {  // (staged in LDS: publish_records ranks from these; in global memory finalize_records derives them itself)
            l.ts_us[idx] = timedelta_us(start_time(a.dp, start));
            l.dur_us[idx] = timedelta_us(run_duration(a.dp, start, end));
        }
    }
}

// ---- cross-lane steps without LDS round trips (a ds_bpermute is ~100 cycles of latency per dependent step) ----
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);
}
// the value lane (lane ^ OFF) holds, OFF = 1, 2, 4, 8 (inside a row of 16 lanes)
template <int OFF>
__device__ __forceinline__ int xor_lane(int v) {
    static_assert(OFF == 1 || OFF == 2 || OFF == 4 || OFF == 8, "row-local offsets only");
    if constexpr (OFF == 1) return dpp_mov<0xB1>(v);                   // quad_perm [1,0,3,2]
    else if constexpr (OFF == 2) return dpp_mov<0x4E>(v);              // quad_perm [2,3,0,1]
    else if constexpr (OFF == 4) return dpp_mov<0x141>(dpp_mov<0x1B>(v));  // quads reversed, then the halves of eight mirrored: i -> i ^ 4
    else return dpp_mov<0x128>(v);                                     // row_ror:8
}
// One step of a butterfly reduction, every lane: op(v of the lane, v of lane ^ OFF) -- the pairs of the halving fold of
// rt::run_stats (p[l] op= p[l + OFF]), so a chain OFF = 32, 16 .. 1 leaves that fold's result in every lane, bit for bit as the
// __shfl_xor chain it replaces did (op is commutative; no re-association).  OFF = 16 / 32 use gfx950's row / half swaps:
// v_permlane16_swap (v, v) yields (rows 0 0 2 2, rows 1 1 3 3) of v, v_permlane32_swap (lower half twice, upper half twice).
template <int OFF, class T, class Op>
__device__ __forceinline__ T butterfly_step(T v, Op op) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "32- or 64-bit values");
    if constexpr (sizeof(T) == 4) {
        const int bits = __builtin_bit_cast(int, v);
        if constexpr (OFF <= 8) {
            return op(v, __builtin_bit_cast(T, xor_lane<OFF>(bits)));
        } else {
            const auto r = (OFF == 16) ? __builtin_amdgcn_permlane16_swap((unsigned)bits, (unsigned)bits, false, false)
                                       : __builtin_amdgcn_permlane32_swap((unsigned)bits, (unsigned)bits, false, false);
            return op(__builtin_bit_cast(T, (int)r[0]), __builtin_bit_cast(T, (int)r[1]));
        }
    } else {
        const unsigned long long bits = __builtin_bit_cast(unsigned long long, v);
        const int lo = (int)(uint32_t)bits, hi = (int)(uint32_t)(bits >> 32);
        if constexpr (OFF <= 8) {
            const unsigned long long o = ((unsigned long long)(uint32_t)xor_lane<OFF>(hi) << 32) | (uint32_t)xor_lane<OFF>(lo);
            return op(v, __builtin_bit_cast(T, o));
        } else {
            const auto rl = (OFF == 16) ? __builtin_amdgcn_permlane16_swap((unsigned)lo, (unsigned)lo, false, false)
                                        : __builtin_amdgcn_permlane32_swap((unsigned)lo, (unsigned)lo, false, false);
            const auto rh = (OFF == 16) ? __builtin_amdgcn_permlane16_swap((unsigned)hi, (unsigned)hi, false, false)
                                        : __builtin_amdgcn_permlane32_swap((unsigned)hi, (unsigned)hi, false, false);
            const unsigned long long x = ((unsigned long long)rh[0] << 32) | rl[0], y = ((unsigned long long)rh[1] << 32) | rl[1];
            return op(__builtin_bit_cast(T, x), __builtin_bit_cast(T, y));
        }
    }
}
template <class T, class Op>
__device__ __forceinline__ T butterfly_all(T v, Op op) {
    v = butterfly_step<32>(v, op);
    v = butterfly_step<16>(v, op);
    v = butterfly_step<8>(v, op);
    v = butterfly_step<4>(v, op);
    v = butterfly_step<2>(v, op);
    v = butterfly_step<1>(v, op);
    return v;
}
// inclusive scan over the 64 lanes of a wave with `op` (associative, identity `ident`): row_shr 1 / 2 / 4 / 8 inside the rows of
// 16, then row_bcast15 / row_bcast31 carry the rows' totals upwards (lanes without a source keep the identity)
template <class Op>
__device__ __forceinline__ int wave_scan_inclusive(int v, int ident, Op op) {
    v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x111, 0xF, 0xF, false));  // row_shr:1
    v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x112, 0xF, 0xF, false));  // row_shr:2
    v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x114, 0xF, 0xF, false));  // row_shr:4
    v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x118, 0xF, 0xF, false));  // row_shr:8
    v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x142, 0xA, 0xF, false));  // row_bcast15 -> rows 1, 3
    v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x143, 0xC, 0xF, false));  // row_bcast31 -> rows 2, 3
    return v;
}

// phase 2: np.max / np.mean / np.std(dB(.)) of one plateau by one wave, in the
// canonical order of rt::run_stats (64 interleaved partials, halving fold).
template <class Cell>
__device__ __forceinline__ RunStats run_stats_wave(int n, Cell cell) {
    const int lane = threadIdx.x & 63;
    double ps = 0.0, pd = 0.0;
    float pm = -INFINITY;
    int any_nan = 0;
    for (int k = lane; k < n; k += 64) {
        const float v = cell(k);
        ps += (double)v;
        pd += (double)db10(v);
        if (v != v) any_nan = 1;
        if (v > pm) pm = v;
    }
    // (butterflies: every lane ends with the fold's result -- the same operations in the same pairs as the host's halving fold)
    const auto add64 = [](double x, double y) { return x + y; };
    ps = butterfly_all(ps, add64);
    pd = butterfly_all(pd, add64);
    pm = butterfly_all(pm, [](float x, float y) { return y > x ? y : x; });
    any_nan = butterfly_all(any_nan, [](int x, int y) { return x | y; });
    const double mean_db = pd / (double)n;
    double pa = 0.0;
    for (int k = lane; k < n; k += 64) {
        const double d = (double)db10(cell(k)) - mean_db;
        pa += d * d;
    }
    pa = butterfly_all(pa, add64);
    RunStats r;
    r.max_p = any_nan ? NAN : pm;
    r.mean_p = (float)(ps / (double)n);
    r.std_db = (float)sqrt(pa / (double)n);
    return r;
}

// The same statistics for 64 / W plateaus at once, W = 32 or 16 lanes each (round 6: a wave that holds a whole stream's lists has a
// dozen or two plateaus to finish, most of them under 32 cells).  A plateau of at most W cells folded over its W lanes IS the 64-lane
// fold of rt::run_stats: the lanes above W hold the identity there (0.0 + x, max(x, -inf), x | 0), so the steps that would add them
// are left out and every group of W lanes runs the remaining steps on its own plateau -- bit for bit run_stats_wave's results.
// n, cell: the lane's own plateau (the same in the W lanes of a group; n = 0: none, the result is not used).
template <int W, class Cell>
__device__ __forceinline__ RunStats run_stats_lanes(int n, Cell cell) {
    static_assert(W == 32 || W == 16, "half or quarter waves");
    const int k = (int)(threadIdx.x & (W - 1));
    const bool has = k < n;
    const float v = has ? cell(k) : 0.f;
    const float dbv = has ? db10(v) : 0.f;
    double ps = has ? 0.0 + (double)v : 0.0, pd = has ? 0.0 + (double)dbv : 0.0;
    float pm = (has && v > -INFINITY) ? v : -INFINITY;
    int any_nan = (has && v != v) ? 1 : 0;
    const auto add64 = [](double x, double y) { return x + y; };
    const auto maxf = [](float x, float y) { return y > x ? y : x; };
    const auto ori = [](int x, int y) { return x | y; };
    if constexpr (W == 32) {
        ps = butterfly_step<16>(ps, add64);
        pd = butterfly_step<16>(pd, add64);
        pm = butterfly_step<16>(pm, maxf);
        any_nan = butterfly_step<16>(any_nan, ori);
    }
    ps = butterfly_step<1>(butterfly_step<2>(butterfly_step<4>(butterfly_step<8>(ps, add64), add64), add64), add64);
    pd = butterfly_step<1>(butterfly_step<2>(butterfly_step<4>(butterfly_step<8>(pd, add64), add64), add64), add64);
    pm = butterfly_step<1>(butterfly_step<2>(butterfly_step<4>(butterfly_step<8>(pm, maxf), maxf), maxf), maxf);
    any_nan = butterfly_step<1>(butterfly_step<2>(butterfly_step<4>(butterfly_step<8>(any_nan, ori), ori), ori), ori);
    const double mean_db = pd / (double)n;
    const double d = (double)dbv - mean_db;
    double pa = has ? 0.0 + d * d : 0.0;
    if constexpr (W == 32) pa = butterfly_step<16>(pa, add64);
    pa = butterfly_step<1>(butterfly_step<2>(butterfly_step<4>(butterfly_step<8>(pa, add64), add64), add64), add64);
    RunStats r;
    r.max_p = any_nan ? NAN : pm;
    r.mean_p = (float)(ps / (double)n);
    r.std_db = (float)sqrt(pa / (double)n);
    return r;
}

// Last step of a call's last kernel (finalize_records / detect_dense), thread 0 of every workgroup:
// the workgroup that takes the last ticket copies the counter words to pinned host memory and
// leaves them zero for the slot's next call -- no reset launch before a call, no copy after it.
__device__ __forceinline__ void close_call(const DetectArgs &a) {
    __threadfence();
    const unsigned long long ticket = atomicAdd(&a.counters[3], 1ull);
    if (ticket + 1 == (unsigned long long)gridDim.x) {
        __threadfence();
        for (int i = 0; i < kCounterWords; ++i) {
            if (i == 3) continue;
            a.host_counters[i] = __hip_atomic_load(&a.counters[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.counters[i], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        a.host_counters[3] = ticket + 1;
        __hip_atomic_store(&a.counters[3], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// order the stream's records by (fi, start), apply the shadow filter against
// the unfiltered list (analyze.py:325) and publish them.
__device__ void publish_records(const DetectArgs &a, RecLds &l, int s, int n) {
    int *lds_base = l.count + 1;
    if (threadIdx.x == 0) {
        long long base = (long long)atomicAdd(&a.counters[0], (unsigned long long)n);
        // a pool too short for this stream: the first records (in (bin, start) order) that still fit are delivered, the
        // call is flagged, and counters[0] tells the host how large a pool the call wants (rt_fetch grows it)
        long long fit = a.pool_cap - base;
        fit = fit < 0 ? 0 : (fit > n ? n : fit);
        if (fit < n) atomicOr(&a.counters[2], kFlagRecOverflow);
        lds_base[0] = fit > 0 ? (int)base : -1;
        lds_base[1] = (int)fit;
        a.rec_offset[s] = fit > 0 ? (int)base : 0;
        a.rec_count[s] = (int)fit;
    }
    __syncthreads();
    const int base = lds_base[0], n_fit = lds_base[1];
    if (base < 0) return;
    const float cal_db = a.cal_s ? a.cal_s[s] : a.dp.cal_db;
    // rt::rank_and_shadow with the dBW figure of every record's maximum computed once (parked in the record's
    // unused `reserved` word) instead of twice per pair: the shadow test is O(n^2) per stream
    for (int i = threadIdx.x; i < n; i += blockDim.x) l.rec[i].reserved = __float_as_int(db10(l.rec[i].max_p) - cal_db);
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int32_t fi = l.rec[i].fi, st = l.rec[i].start;
        const float mx_i = __int_as_float(l.rec[i].reserved);
        const long long ts_i = l.ts_us[i], dur_i = l.dur_us[i];
        int rank = 0, shadow = 0;
        for (int j = 0; j < n; ++j) {
            const rt_record &rj = l.rec[j];
            if (rj.fi < fi || (rj.fi == fi && rj.start < st)) ++rank;
            if (shadowed_by(ts_i, dur_i, mx_i, l.ts_us[j], l.dur_us[j], __int_as_float(rj.reserved))) shadow = 1;
        }
        rt_record out = l.rec[i];
        out.shadowed = shadow;
        out.reserved = 0;
        if (rank < n_fit) a.records[(int64_t)base + rank] = out;
    }
}

// clamp the candidate count after phase 1 (all threads), flag truncation
__device__ __forceinline__ int settled_count(const DetectArgs &a, RecLds &l) {
    __syncthreads();
    int n = *l.count;
    if (n > a.rec_cap) {
        if (threadIdx.x == 0) {
            atomicOr(&a.counters[2], kFlagRecOverflow);
            atomicMax(&a.counters[4], (unsigned long long)n);  // (the capacity this stream wants: rt_fetch grows the handle's and analyses the call again)
        }
        n = a.rec_cap;
    }
    return n;
}

// ---------------------------------------------------------------------------
// sparse detection: one WAVE per (stream, bucket), bucket = bin & (kBuckets-1)
// ---------------------------------------------------------------------------
constexpr int kSmallBucket = 1024;  // buckets up to this many cells use the small-LDS instantiation
constexpr int kQuarters = 4;        // detect_group: a wave takes a stream's buckets all together or a quarter of them
constexpr int kCandCapMax = 64;     // plateaus a bucket wave stages in LDS before it finishes them (a.cand_cap <= this); no limit per bucket

// Bitonic sort of 64*M (key, value) pairs held in registers, element i = m*64 + lane.
// Compare-exchange distances >= 64 pair two registers of the same lane (no data
// movement), distances < 64 pair the same register of two lanes (ds_bpermute via
// __shfl_xor): no LDS memory, no synchronisation.
template <int M>
__device__ __forceinline__ void wave_bitonic_sort(uint32_t (&k)[M], float (&v)[M], int lane) {
    constexpr int N2 = 64 * M;
#pragma unroll
    for (int kk = 2; kk <= N2; kk <<= 1) {
#pragma unroll
        for (int j = kk >> 1; j >= 64; j >>= 1) {  // register <-> register steps
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const int pm = m ^ (j / 64);
                if (pm > m) {
                    const bool up = (((m * 64) & kk) == 0);
                    const bool sw = up ? (k[m] > k[pm]) : (k[m] < k[pm]);
                    const uint32_t a = k[m], b = k[pm];
                    const float x = v[m], y = v[pm];
                    k[m] = sw ? b : a;
                    k[pm] = sw ? a : b;
                    v[m] = sw ? y : x;
                    v[pm] = sw ? x : y;
                }
            }
        }
#pragma nounroll
        for (int j = (kk >> 1) < 32 ? (kk >> 1) : 32; j > 0; j >>= 1) {  // lane <-> lane steps
            const bool lower = (lane & j) == 0;
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const bool up = (((m * 64 + lane) & kk) == 0);
                const uint32_t ok = (uint32_t)__shfl_xor((int)k[m], j, 64);
                const float ov = __shfl_xor(v[m], j, 64);
                const bool take = (lower == up) ? (ok < k[m]) : (ok > k[m]);
                k[m] = take ? ok : k[m];
                v[m] = take ? ov : v[m];
            }
        }
    }
}

// load a bucket's cells into registers, sort, leave them in LDS in (bin, t) order
template <int M>
__device__ __forceinline__ void sort_bucket_regs(const uint2 *src, int n, int lane, uint32_t *keys, float *vals) {
    uint32_t k[M];
    float v[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const int i = m * 64 + lane;
        const uint2 e = (i < n) ? src[i] : make_uint2(0xFFFFFFFFu, 0u);
        k[m] = e.x;
        v[m] = __uint_as_float(e.y);
    }
    wave_bitonic_sort<M>(k, v, lane);
#pragma unroll
    for (int m = 0; m < M; ++m) {
        keys[m * 64 + lane] = k[m];
        vals[m * 64 + lane] = v[m];
    }
}

// v_min_u32 / v_max_u32 spelled as opaque instructions: hipcc's value tracking through a fully unrolled
// min/max network (55 stages x 16 registers) takes tens of minutes of compile time otherwise
__device__ __forceinline__ uint32_t umin_op(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_min_u32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t umax_op(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_max_u32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// The same network on single 32-bit words (min / max per compare-exchange, one shuffle instead of two).
template <int M>
__device__ __forceinline__ void wave_bitonic_sort_u32(uint32_t (&k)[M], int lane) {
    constexpr int N2 = 64 * M;
#pragma unroll
    for (int kk = 2; kk <= N2; kk <<= 1) {
#pragma unroll
        for (int j = kk >> 1; j >= 64; j >>= 1) {
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const int pm = m ^ (j / 64);
                if (pm > m) {
                    const bool up = (((m * 64) & kk) == 0);
                    const uint32_t lo = umin_op(k[m], k[pm]);
                    const uint32_t hi = umax_op(k[m], k[pm]);
                    k[m] = up ? lo : hi;
                    k[pm] = up ? hi : lo;
                }
            }
        }
#pragma nounroll
        for (int j = (kk >> 1) < 32 ? (kk >> 1) : 32; j > 0; j >>= 1) {
            const bool lower = (lane & j) == 0;
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const bool up = (((m * 64 + lane) & kk) == 0);
                const uint32_t o = (uint32_t)__shfl_xor((int)k[m], j, 64);
                const uint32_t lo = umin_op(o, k[m]);
                const uint32_t hi = umax_op(o, k[m]);
                k[m] = (lower == up) ? lo : hi;
            }
        }
    }
}

// Bucket sort with one word per cell: all bins of a bucket share their low log2(kBuckets) bits, so
// (bin / kBuckets, t) and the cell's position in the unsorted list (< 1024) fit 32 bits together
// (`shift` = tbits, checked by the caller).  The powers wait in LDS (in the `keys` area) and are
// gathered by position after the sort -- 2.5x fewer instructions than sorting (key, value) pairs.
template <int M>
__device__ __forceinline__ void sort_bucket_packed(const uint2 *src, int n, int lane, uint32_t *keys, float *vals,
                                                   int tbits, int bkt) {
    constexpr int kIdxBits = 10;
    const uint32_t tmask = (1u << tbits) - 1u;
    uint32_t k[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const int i = m * 64 + lane;
        uint32_t word = 0xFFFFFFFFu;
        if (i < n) {
            const uint2 e = src[i];
            const uint32_t hi = (e.x >> tbits) / kBuckets;
            word = (((hi << tbits) | (e.x & tmask)) << kIdxBits) | (uint32_t)i;
            keys[i] = e.y;  // the power, parked until the order is known
        }
        k[m] = word;
    }
    wave_sync();
    wave_bitonic_sort_u32<M>(k, lane);
    float v[M];
#pragma unroll
    for (int m = 0; m < M; ++m) v[m] = (m * 64 + lane < n) ? __uint_as_float(keys[k[m] & ((1u << kIdxBits) - 1u)]) : 0.f;
    wave_sync();
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const int i = m * 64 + lane;
        const uint32_t w = k[m] >> kIdxBits;
        keys[i] = (i < n) ? (((((w >> tbits) * kBuckets) | (uint32_t)bkt) << tbits) | (w & tmask)) : 0xFFFFFFFFu;
        vals[i] = v[m];
    }
}

// Where a bucket's key space (bin in bucket, time) has at most 2^15 values -- nperseg 256 with up to 2048 segments per
// buffer, the reference's default geometry and BASELINE config 4 -- the order comes from a bitmap instead of a sorting
// network: every cell sets its bit (32768 bits = the wave's `keys` area), a prefix count over the 1024 words (16 per
// lane, one wave scan; parked in the `vals` area) gives every cell its rank, and the cells, held in registers meanwhile,
// go to their places.  ~250 instructions whatever the size, against ~1100 for the network at 256 cells.
template <int M>
__device__ __forceinline__ void sort_bucket_bitmap(const uint2 *src, int n, int lane, uint32_t *keys, float *vals,
                                                   int tbits, int bkt) {
    const uint32_t tmask = (1u << tbits) - 1u;
    uint32_t ck[M], cv[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const int i = m * 64 + lane;
        ck[m] = 0xFFFFFFFFu;
        cv[m] = 0u;
        if (i < n) {
            const uint2 e = src[i];
            ck[m] = (((e.x >> tbits) / kBuckets) << tbits) | (e.x & tmask);
            cv[m] = e.y;
        }
    }
    uint4 *const kw4 = reinterpret_cast<uint4 *>(keys);
    uint4 *const pv4 = reinterpret_cast<uint4 *>(vals);
    uint32_t *const pre = reinterpret_cast<uint32_t *>(vals);
#pragma unroll
    for (int j = 0; j < 4; ++j) kw4[lane * 4 + j] = make_uint4(0u, 0u, 0u, 0u);
    wave_sync();
#pragma unroll
    for (int m = 0; m < M; ++m)
        if (ck[m] != 0xFFFFFFFFu) atomicOr(&keys[ck[m] >> 5], 1u << (ck[m] & 31u));
    wave_sync();
    uint32_t run = 0;
    uint4 before[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint4 w = kw4[lane * 4 + j];
        before[j].x = run;
        run += (uint32_t)__builtin_popcount(w.x);
        before[j].y = run;
        run += (uint32_t)__builtin_popcount(w.y);
        before[j].z = run;
        run += (uint32_t)__builtin_popcount(w.z);
        before[j].w = run;
        run += (uint32_t)__builtin_popcount(w.w);
    }
    const uint32_t incl = (uint32_t)wave_scan_inclusive((int)run, 0, [](int x, int y) { return x + y; });
    const uint32_t excl = incl - run;
#pragma unroll
    for (int j = 0; j < 4; ++j) pv4[lane * 4 + j] = make_uint4(before[j].x + excl, before[j].y + excl, before[j].z + excl, before[j].w + excl);
    wave_sync();
    uint32_t rk[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        rk[m] = 0u;
        if (ck[m] != 0xFFFFFFFFu) {
            const uint32_t w = ck[m] >> 5;
            rk[m] = pre[w] + (uint32_t)__builtin_popcount(keys[w] & ((1u << (ck[m] & 31u)) - 1u));
        }
    }
    wave_sync();
#pragma unroll
    for (int m = 0; m < M; ++m) {
        if (ck[m] != 0xFFFFFFFFu) {
            keys[rk[m]] = ((((ck[m] >> tbits) * kBuckets) | (uint32_t)bkt) << tbits) | (ck[m] & tmask);
            vals[rk[m]] = __uint_as_float(cv[m]);
        }
    }
    for (int i = n + lane; i < 64 * M; i += 64) {
        keys[i] = 0xFFFFFFFFu;
        vals[i] = 0.f;
    }
}

#ifndef RT_DETECT_ABLATE
#define RT_DETECT_ABLATE 0  // diagnostic builds only: 1 = no row means, 2 = no sort, 3 = stop after the sort, 4 = no run statistics, 5 = no hand-over of the records, 6 = stop before the runs are gated, 7 = (detect_group) stop after the row means, 8 = (finalize_records) no per-stream words to host memory, 9 = (finalize_records) records to device memory instead of the host pool
#endif
// The second half of a bucket wave's work, on a list that lies in LDS in (bin, t) order (keys = bin << tbits | t, vals = the cells'
// powers; avg_of(bin) = the bin's row mean): predicate -> maximal runs -> gates -> statistics -> raw records of stream s.
// Shared by the wave of one bucket (detect_bucket_one) and the wave of a group of buckets (detect_group).
template <class AvgOf>
__device__ __forceinline__ void detect_runs(const DetectArgs &a, const DetectParams &dp, const int s, const int n, uint32_t *keys, const float *vals,
                                            rt_record *cand, const int lane, AvgOf avg_of) {
    const int F = a.n_bins;
    const int T = a.dp.n_seg;
    const uint32_t tmask = (1u << a.tbits) - 1u;
    // the predicate (analyze.py:370, 378) is evaluated where it is needed: one float division
    auto is_above = [&](int i) -> bool {
        const int bin = (int)(keys[i] >> a.tbits);
        return cell_above(vals[i], avg_of(bin), dp.thr, dp.snr);
    };

    // Statistics of the staged plateaus (the whole wave per plateau, canonical order of rt::run_stats), then
    // hand them to the stream's unordered list (finalize_records orders and filters).  Called whenever the
    // staging area is full and once at the end: a bucket may hold any number of plateaus (dense tag trains at
    // nperseg 4096 put hundreds into one), only the stream's record_capacity limits them.
    auto drain = [&](int ncand) {
        wave_sync();
        for (int c = 0; c < ncand;) {
            // the next plateaus four at a time on quarter waves (all of them <= 16 cells), two on half waves (<= 32), or one on the wave
            const int left = ncand - c;
            const auto len_of = [&](int j) -> int { return __builtin_amdgcn_readfirstlane(cand[c + j].end - cand[c + j].start); };
            int per = 1;
            if (left >= 2) {
                const int l01 = max(len_of(0), len_of(1));
                if (l01 <= 32) per = 2;
                if (left >= 4 && l01 <= 16 && max(len_of(2), len_of(3)) <= 16) per = 4;
            }
            const int W = 64 / per;
            const int g = lane / W;  // the lane's plateau of this pass
            const int cm = c + g;
            const int start = cand[cm].start, off = cand[cm].reserved, fi = cand[cm].fi, end = cand[cm].end;
            PrevCells prev{a.prev + ((int64_t)s * a.prev_cols + a.prev_cols) * F + fi, F};
            auto cell = [&](int k) -> float {
                const int t = start + k;
                return t < 0 ? prev(-t) : vals[off + t];
            };
            const RunStats st = (RT_DETECT_ABLATE == 4) ? RunStats{cell(0), cell(1), 0.f}
                                : per == 4          ? run_stats_lanes<16>(end - start, cell)
                                : per == 2          ? run_stats_lanes<32>(end - start, cell)
                                                    : run_stats_wave(end - start, cell);
            if ((lane & (W - 1)) == 0) {
                cand[cm].max_p = st.max_p;
                cand[cm].mean_p = st.mean_p;
                cand[cm].std_db = st.std_db;
                cand[cm].reserved = 0;
            }
            c += per;
        }
        wave_sync();
        if (RT_DETECT_ABLATE == 5) return;
        int slot = 0;
        if (lane == 0) slot = atomicAdd(&a.raw_count[s], ncand);
        slot = __builtin_amdgcn_readfirstlane(slot);
        for (int c = lane; c < ncand; c += 64) {
            if (slot + c < a.rec_cap)
                a.raw[(int64_t)s * a.rec_cap + slot + c] = cand[c];
            else
                atomicOr(&a.counters[2], kFlagRecOverflow);
        }
        wave_sync();  // the staging area is free again
    };

    // maximal runs of above-cells: a run's last cell learns the index of its first cell from
    // an inclusive prefix-max over "index if run start else -1" (64 cells per step + carry)
    int ncand = 0;       // wave-uniform
    int carry = -1;      // wave-uniform: last run start seen in earlier steps
    for (int base_i = 0; base_i < n; base_i += 64) {
        const int i = base_i + lane;
        const bool valid = i < n;
        const uint32_t key = valid ? keys[i] : 0u;
        const int t = (int)(key & tmask);
        const bool ab = valid && is_above(i);
        const bool prev_adj = ab && i > 0 && t > 0 && keys[i - 1] == key - 1 && is_above(i - 1);
        const bool next_adj = ab && (i + 1 < n) && (t + 1 < T) && keys[i + 1] == key + 1 && is_above(i + 1);
        const bool is_start = ab && !prev_adj;
        const bool is_end = ab && !next_adj;
        int first = wave_scan_inclusive(is_start ? i : -1, -1, [](int x, int y) { return x > y ? x : y; });
        first = first > carry ? first : carry;
        carry = __builtin_amdgcn_readlane(first, 63);

        bool keep = false;
        int fi = 0, b = 0, e = 0, start = 0;
        float av = 0.f;
        if (RT_DETECT_ABLATE == 6) { if (__builtin_amdgcn_ballot_w64(is_end) == 12345ull) keys[0] = 1u; continue; }
        if (is_end) {
            const uint32_t key0 = keys[first];
            fi = (int)(key >> a.tbits);
            b = (int)(key0 & tmask);
            e = t + 1;
            if (b > 0 && (first == 0 || keys[first - 1] != key0 - 1)) {
                // the cell before a run must have been emitted by the scan (T11)
                if (!a.filtered) {
                    atomicOr(&a.counters[2], kFlagInconsistent);
                    if (a.stream_incons) a.stream_incons[s] = 1;
                }
            } else {
                av = avg_of(fi);
                PrevCells prev{a.prev + ((int64_t)s * a.prev_cols + a.prev_cols) * F + fi, F};
                keep = gate_run(dp, b, e, av, prev, &start);
            }
        }
        // stage the gated plateaus; when the staging area runs full, the staged ones are finished first
        unsigned long long todo = __builtin_amdgcn_ballot_w64(keep);
        while (todo) {
            if (ncand == a.cand_cap) {
                drain(ncand);
                ncand = 0;
            }
            const int room = a.cand_cap - ncand;
            const bool mine = keep && ((todo >> lane) & 1ull);
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(todo >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)todo, 0));
            const bool now = mine && rank < room;
            if (now) {
                rt_record r;
                r.stream = s;
                r.fi = fi;
                r.start = start;
                r.end = e;
                r.max_p = 0.f;
                r.mean_p = 0.f;
                r.std_db = 0.f;
                r.row_mean = av;
                r.shadowed = 0;
                r.reserved = first - b;  // cell t of this bin sits at vals[reserved + t]
                cand[ncand + rank] = r;
            }
            const unsigned long long done = __builtin_amdgcn_ballot_w64(now);
            ncand += __builtin_popcountll(done);
            todo &= ~done;
        }
    }
    if (ncand) drain(ncand);
}


// All cells of a bin live in one bucket, so a wave can finish its bins alone:
// row means -> sort by (bin, t) -> predicate -> maximal runs (paired by a
// prefix-max scan, no sequential walks) -> gates -> statistics -> raw records.
// LARGE = false handles buckets of <= kSmallBucket cells (static 10 KiB of
// LDS, many waves per CU, co-resident with the scan); LARGE = true handles the
// rare bigger ones (up to hot_cap cells, dynamic LDS).  Both are launched; a
// wave whose bucket belongs to the other instantiation exits at once.
template <bool LARGE>
__device__ __forceinline__ void detect_bucket_one(const DetectArgs &a, const int sb, const int wave, const int lane, unsigned char *dyn_smem) {
    if (sb >= a.n_streams * kBuckets) return;
    const int s = sb / kBuckets, bkt = sb % kBuckets;
    const int F = a.n_bins;
    const int T = a.dp.n_seg;
    const uint32_t n_raw = a.hot_count[sb];  // (finalize_records, the call's last kernel, sums the counters per stream and leaves them zero)
    if (!LARGE && lane == 0 && n_raw > (uint32_t)kSmallBucket && n_raw <= (uint32_t)a.hot_cap) a.large_any[s] = 1u;  // the large instantiation has work in this stream
    if (n_raw == 0) return;
    if (n_raw > (uint32_t)a.hot_cap) {
        if (!LARGE && lane == 0) {
            atomicOr(&a.counters[2], kFlagHotOverflow);
            if (a.stream_overflow) a.stream_overflow[s] = 1;
        }
        return;
    }
    if (LARGE != (n_raw > (uint32_t)kSmallBucket)) return;
    const int n = (int)n_raw;
    int n2 = 64;
    while (n2 < n) n2 <<= 1;

    // per-wave LDS: keys | vals | row means of the bucket's bins | plateau candidates
    const int cap2 = LARGE ? a.lds_cells : kSmallBucket;
    const int nbins_b = F / kBuckets;
    const size_t wave_bytes = (size_t)cap2 * 8 + (((size_t)nbins_b * 4 + 15) & ~(size_t)15) + sizeof(rt_record) * a.cand_cap;
    unsigned char *base = dyn_smem + (LARGE ? 0 : (size_t)wave * wave_bytes);
    uint32_t *keys = reinterpret_cast<uint32_t *>(base);
    float *vals = reinterpret_cast<float *>(keys + cap2);
    float *avg = vals + cap2;
    rt_record *cand = reinterpret_cast<rt_record *>(base + (size_t)cap2 * 8 + (((size_t)nbins_b * 4 + 15) & ~(size_t)15));
    const uint32_t tmask = (1u << a.tbits) - 1u;
    const DetectParams dp = stream_params(a, s);

    // row means of the bucket's bins: np.mean(row) (analyze.py:375) from the scan's partial sums
    // Where a bucket holds many bins (nperseg >= 512: 32 ... 256 of them, each a sum over the stream's partial rows -- 11 at BASELINE
    // config 5, 19 at config 3) only the bins that OCCUR in the list get their mean, after the sort below (a dozen of 256 at config 5,
    // where the full table was four dependent rounds of loads per lane: detect_bucket 310 -> 228 us on the config-5 share, 345 -> 305
    // at config 3, profiles/r05_h_*).  The same sums, bit for bit.
    const bool lazy_means = !LARGE && F / kBuckets > 16;
    for (int r = lane; r < F / kBuckets && (!LARGE || wave == 0) && !lazy_means; r += 64) {
        const int bin = bkt + kBuckets * r;
        if (RT_DETECT_ABLATE == 1) { avg[r] = 1e-20f; continue; }
        avg[r] = (float)row_sum_from_partials(a.psum + (int64_t)s * a.chunks * F + bin, a.chunks, F) / (float)T;
    }
    const uint2 *src = a.hot + (int64_t)sb * a.hot_cap;
    if constexpr (!LARGE) {
        // <= 1024 cells: sort in registers
        int hb = 0;  // bits of bin / kBuckets
        while ((1 << hb) < F / kBuckets) ++hb;
        const bool packed = hb + a.tbits + 10 <= 32;
        const bool bitmap = hb + a.tbits <= 15 && n2 >= 128;  // (64 cells: the network is as cheap)
        switch (RT_DETECT_ABLATE == 2 ? 0 : (bitmap ? n2 + 1 : packed ? n2 : -n2)) {
            case 0: for (int i = lane; i < n; i += 64) { keys[i] = src[i].x; vals[i] = __uint_as_float(src[i].y); } break;
            case 129: sort_bucket_bitmap<2>(src, n, lane, keys, vals, a.tbits, bkt); break;
            case 257: sort_bucket_bitmap<4>(src, n, lane, keys, vals, a.tbits, bkt); break;
            case 513: sort_bucket_bitmap<8>(src, n, lane, keys, vals, a.tbits, bkt); break;
            case 1025: sort_bucket_bitmap<16>(src, n, lane, keys, vals, a.tbits, bkt); break;
            case 64: sort_bucket_packed<1>(src, n, lane, keys, vals, a.tbits, bkt); break;
            case 128: sort_bucket_packed<2>(src, n, lane, keys, vals, a.tbits, bkt); break;
            case 256: sort_bucket_packed<4>(src, n, lane, keys, vals, a.tbits, bkt); break;
            case 512: sort_bucket_packed<8>(src, n, lane, keys, vals, a.tbits, bkt); break;
            case 1024: sort_bucket_packed<16>(src, n, lane, keys, vals, a.tbits, bkt); break;
            // very long buffers (bin and time bits leave no room for the position): (key, value) pairs
            case -64: sort_bucket_regs<1>(src, n, lane, keys, vals); break;
            case -128: sort_bucket_regs<2>(src, n, lane, keys, vals); break;
            case -256: sort_bucket_regs<4>(src, n, lane, keys, vals); break;
            case -512: sort_bucket_regs<8>(src, n, lane, keys, vals); break;
            default: sort_bucket_regs<16>(src, n, lane, keys, vals); break;
        }
        wave_sync();
        if (RT_DETECT_ABLATE == 3) return;
        if (lazy_means) {
            // the list is in (bin, t) order: the first cell of every bin computes that bin's mean -- all of a step's bins at once
            for (int base_i = 0; base_i < n; base_i += 64) {
                const int i = base_i + lane;
                if (i < n) {
                    const int bin = (int)(keys[i] >> a.tbits);
                    if (i == 0 || (int)(keys[i - 1] >> a.tbits) != bin)
                        avg[bin / kBuckets] = (RT_DETECT_ABLATE == 1) ? 1e-20f
                                                                      : (float)row_sum_from_partials(a.psum + (int64_t)s * a.chunks * F + bin, a.chunks, F) / (float)T;
                }
            }
            wave_sync();
        }
    } else {
        const int tid = threadIdx.x;
        for (int i = tid; i < n2; i += 256) {
            if (i < n) {
                const uint2 e = src[i];
                keys[i] = e.x;
                vals[i] = __uint_as_float(e.y);
            } else {
                keys[i] = 0xFFFFFFFFu;
                vals[i] = 0.f;
            }
        }
        __syncthreads();
        // bitonic sort in LDS by key = (bin, t), all 256 threads; keys are unique (one entry per cell)
        for (int k = 2; k <= n2; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < (n2 >> 1); t += 256) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    const int ixj = i | j;
                    const bool up = ((i & k) == 0);
                    const uint32_t ki = keys[i], kj = keys[ixj];
                    if ((ki > kj) == up) {
                        keys[i] = kj;
                        keys[ixj] = ki;
                        const float tv = vals[i];
                        vals[i] = vals[ixj];
                        vals[ixj] = tv;
                    }
                }
                __syncthreads();
            }
        }
        if (wave != 0) return;  // no block-wide barrier below this point
    }

    detect_runs(a, dp, s, n, keys, vals, cand, lane, [&](int bin) -> float { return avg[bin / kBuckets]; });
}

template <bool LARGE>
__global__ __launch_bounds__(256) void detect_bucket(const DetectArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
    // small: 4 waves = 4 buckets per workgroup; large: the 4 waves sort ONE bucket together in
    // LDS, then wave 0 finishes it alone
    // wave index made provably uniform; lane id from mbcnt (hipcc's value tracking on
    // `threadIdx.x & 63` sends the unrolled register sort into a compile-time blow-up)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if constexpr (!LARGE) {
        detect_bucket_one<false>(a, (int)blockIdx.x * 4 + wave, wave, lane, dyn_smem);
    } else {
        // one workgroup per STREAM; it walks the stream's 16 buckets only where the small instantiation has seen a
        // large one.  (One workgroup per bucket, each reading its counter, keeping it for the statistics and zeroing it,
        // cost 0.45 - 0.5 ms per call at 4 096 streams for doing almost nothing; that bookkeeping is finalize_records' now.)
        if ((int)blockIdx.x >= a.n_streams || a.large_any[blockIdx.x] == 0u) return;  // (set by the small instantiation, which ran before)
        for (int b = 0; b < kBuckets; ++b) {
            detect_bucket_one<true>(a, (int)blockIdx.x * kBuckets + b, wave, lane, dyn_smem);
            __syncthreads();  // the next bucket reuses the workgroup's LDS
        }
    }
}

// ---------------------------------------------------------------------------
// sparse detection of a WHOLE STREAM by one wave (round 6).  With thousands of streams per launch and a few hundred candidate cells
// per stream -- the reference's defaults -- a launch of one wave per (stream, list) is tens of thousands of waves of a dozen or two
// cells each, every one paying the same chain of dependent round trips (counter -> list -> row sums -> record slot), sixteen rounds
// of them per launch.  Here a wave takes all sixteen lists of its stream where they hold <= kGroupCells cells together: gathered
// in ONE round trip (a cell's list found from the prefix of the counters), sorted together by (bin, t) -- the key carries the whole
// bin -- and handed to the same detect_runs, which finishes the stream's plateaus two or four at a time.  Streams that hold more go
// to a work list and to the per-list waves behind this launch (detect_bucket_listed), which also own the large lists and the overflow
// flags: a stream that fits has neither.  Same records as the per-list form (the order of a stream's unordered list differs;
// finalize_records orders it).  The host launches this form while the batch's streams are light on average (rt_analyze.hip).
// (Measured and dropped: quarters of a stream's lists per wave for heavier streams -- BASELINE config 4, 1 500 cells per stream:
// 3.76 against 3.16 ms of detection per step; profiles/r06_k_*.)
// LDS of a wave, inside the small-bucket wave's 8 KiB: keys[kGroupCells] | vals[kGroupCells] | row means of up to 256 bins.
// ---------------------------------------------------------------------------
constexpr int kGroupCells = 896;
constexpr int kGroupBins = 256;  // (kSmallBucket - kGroupCells) * 8 / 4

// gather + sort: cell i of the group (i = m*64 + lane) is cell i - start(b) of bucket b, b = the number of buckets whose inclusive
// prefix `ends[.]` is <= i; one word per cell ((bin, t) above the cell's position, as in sort_bucket_packed), the powers parked in LDS
template <int M>
__device__ __forceinline__ void sort_group_packed(const uint2 *hot_s, const int hot_cap, const uint32_t (&ends)[kBuckets], const int n, const int lane,
                                                  uint32_t *keys, float *vals) {
    constexpr int kIdxBits = 10;
    uint32_t k[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const uint32_t i = (uint32_t)(m * 64 + lane);
        uint32_t b = 0u, st = 0u;
#pragma unroll
        for (int j = 0; j < kBuckets; ++j) {
            const bool ge = i >= ends[j];
            b += ge ? 1u : 0u;
            st = ge ? ends[j] : st;
        }
        uint32_t word = 0xFFFFFFFFu;
        if (i < (uint32_t)n) {
            const uint2 e = hot_s[(int64_t)b * hot_cap + (i - st)];
            word = (e.x << kIdxBits) | i;
            keys[i] = e.y;  // the power, parked until the order is known
        }
        k[m] = word;
    }
    wave_sync();
    wave_bitonic_sort_u32<M>(k, lane);
    float v[M];
#pragma unroll
    for (int m = 0; m < M; ++m) v[m] = (m * 64 + lane < n) ? __uint_as_float(keys[k[m] & ((1u << kIdxBits) - 1u)]) : 0.f;
    wave_sync();
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const int i = m * 64 + lane;
        if (i < n) {  // (nothing beyond n is read below; the areas behind kGroupCells belong to the row means)
            keys[i] = k[m] >> kIdxBits;
            vals[i] = v[m];
        }
    }
}

__global__ __launch_bounds__(256) void detect_group(const DetectArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int s = (int)blockIdx.x * 4 + wave;
    if (s >= a.n_streams) return;
    constexpr int QB = kBuckets / kQuarters;
    const uint32_t cnt = lane < kBuckets ? a.hot_count[s * kBuckets + lane] : 0u;
    uint32_t tot_all = 0u;
    bool over_all = false;  // a list that overflowed (its counter says more than the list holds) is not read here
#pragma unroll
    for (int b = 0; b < kBuckets; ++b) {
        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cnt, b);
        tot_all += c;
        over_all |= c > (uint32_t)a.hot_cap;
    }
    const bool whole = tot_all <= (uint32_t)kGroupCells && !over_all;
    if (!whole) {
        // (this stream's lists one by one, behind this launch: detect_bucket_listed; large lists and overflowed ones are in here)
        if (lane < kQuarters) {
            const int slot = atomicAdd(a.work_count, 1);
            a.work_list[slot] = s * kQuarters + lane;
        }
        return;
    }
    (void)QB;
    const int n = (int)tot_all;
    if (n == 0) return;
    // inclusive prefix of the group's counters (buckets outside the group count as empty): scalars
    uint32_t ends[kBuckets];
    {
        uint32_t run = 0u;
#pragma unroll
        for (int b = 0; b < kBuckets; ++b) {
            const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cnt, b);
            run += c;
            ends[b] = run;
        }
    }
    const int F = a.n_bins;
    const int T = a.dp.n_seg;
    const int nbins_b = F / kBuckets;
    const size_t wave_bytes = (size_t)kSmallBucket * 8 + (((size_t)nbins_b * 4 + 15) & ~(size_t)15) + sizeof(rt_record) * a.cand_cap;  // = detect_bucket<false>'s
    unsigned char *base = dyn_smem + (size_t)wave * wave_bytes;
    uint32_t *keys = reinterpret_cast<uint32_t *>(base);
    float *vals = reinterpret_cast<float *>(keys + kGroupCells);
    float *avg = vals + kGroupCells;  // [F], F <= kGroupBins
    rt_record *cand = reinterpret_cast<rt_record *>(base + (size_t)kSmallBucket * 8 + (((size_t)nbins_b * 4 + 15) & ~(size_t)15));
    const DetectParams dp = stream_params(a, s);
    const uint2 *hot_s = a.hot + (int64_t)s * kBuckets * a.hot_cap;
    int n2 = 64;
    while (n2 < n) n2 <<= 1;
    switch (n2) {
        case 64: sort_group_packed<1>(hot_s, a.hot_cap, ends, n, lane, keys, vals); break;
        case 128: sort_group_packed<2>(hot_s, a.hot_cap, ends, n, lane, keys, vals); break;
        case 256: sort_group_packed<4>(hot_s, a.hot_cap, ends, n, lane, keys, vals); break;
        case 512: sort_group_packed<8>(hot_s, a.hot_cap, ends, n, lane, keys, vals); break;
        default: sort_group_packed<16>(hot_s, a.hot_cap, ends, n, lane, keys, vals); break;
    }
    wave_sync();
    if (RT_DETECT_ABLATE == 3) return;
    // row means of the bins that occur: the first cell of every bin computes its bin's (the list is in (bin, t) order)
    for (int base_i = 0; base_i < n; base_i += 64) {
        const int i = base_i + lane;
        if (i < n) {
            const int bin = (int)(keys[i] >> a.tbits);
            if (i == 0 || (int)(keys[i - 1] >> a.tbits) != bin)
                avg[bin] = (float)row_sum_from_partials(a.psum + (int64_t)s * a.chunks * F + bin, a.chunks, F) / (float)T;
        }
    }
    wave_sync();
    if (RT_DETECT_ABLATE == 7) return;
    detect_runs(a, dp, s, n, keys, vals, cand, lane, [&](int bin) -> float { return avg[bin]; });
}

// Behind detect_group: the per-bucket waves for the quarters it left (their four buckets to the four waves of a workgroup), a grid
// that loops over the work list -- empty as a rule.  (A kernel of its own: the loop around the inlined bucket wave costs registers
// -- 137 against 98 -- and with them the fourth wave per SIMD of detect_bucket<false>.)
__global__ __launch_bounds__(256) void detect_bucket_listed(const DetectArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int n_work = __builtin_amdgcn_readfirstlane(*a.work_count);
    for (int w = (int)blockIdx.x; w < n_work; w += (int)gridDim.x) {
        const int e = __builtin_amdgcn_readfirstlane(a.work_list[w]);
        detect_bucket_one<false>(a, (e / kQuarters) * kBuckets + (e % kQuarters) * (kBuckets / kQuarters) + wave, wave, lane, dyn_smem);
        wave_sync();  // the next bucket reuses the wave's LDS
    }
}

// One workgroup per stream: order the stream's records by (fi, start), shadow verdicts, publish.
// (rt::rank_and_shadow against tiles of 256 records in 16 KiB of static LDS: with room for `rec_cap` records -- 64 KiB at
// the default -- two workgroups fit a CU, and a batch of thousands of streams with a dozen records each spent most of this
// kernel waiting for a slot.)  ONE atomic per workgroup on the call's counter word: records allocated in the low 40 bits,
// workgroups arrived above them (two atomics on one address cost thousands of streams ~11 ns each, one after the other).
// The workgroup that arrives last closes the call (close_call's job in detect_dense): the others have made their only
// update of the counters by then; the pool overflow is told from the total.
constexpr int kFinalTile = 256;
constexpr int kTicketShift = 40;
__global__ __launch_bounds__(256) void finalize_records(const DetectArgs a) {
    __shared__ rt_record t_rec[kFinalTile];
    __shared__ long long t_ts[kFinalTile], t_dur[kFinalTile];
    __shared__ long long sh_base;
    __shared__ int sh_fit;
    const int s = a.stream_list ? a.stream_list[blockIdx.x] : (int)blockIdx.x;  // (a list: the dense re-run of a few streams, detect_dense<true>)
    const int tid = threadIdx.x;
    int n = a.raw_count[s];
    // last reader of the stream's sixteen candidate counters: their sum for the statistics, then zero for the slot's next call --
    // sixteen lanes, one round trip (round 6: a loop in thread 0 was sixteen DEPENDENT round trips, the counters being read and
    // written through the same array: 62 us per 4 096 streams before the first record was touched)
    uint32_t hot_sum = 0u;
    if (!a.stream_list && tid < 64) {
        uint32_t c = 0u;
        if (tid < kBuckets) {
            c = a.hot_count[s * kBuckets + tid];
            if (c) a.hot_count_rw[s * kBuckets + tid] = 0u;
        }
        const auto addu = [](uint32_t x, uint32_t y) { return x + y; };
        hot_sum = butterfly_step<1>(butterfly_step<2>(butterfly_step<4>(butterfly_step<8>(c, addu), addu), addu), addu);  // (the row of sixteen lanes)
    }
    const int wanted = n;
    if (n > a.rec_cap) n = a.rec_cap;  // overflow already flagged by the producer
    if (n < 0) n = 0;
    __syncthreads();
    if (tid == 0) {
        if (wanted) a.raw_count[s] = 0;  // ready for the slot's next call
        if (wanted > a.rec_cap) {
            // the stream found more records than the handle's capacity holds: rt_fetch grows it and analyses the call again
            atomicOr(&a.counters[2], kFlagRecOverflow);
            atomicMax(&a.counters[4], (unsigned long long)wanted);
        }
        if (!a.stream_list) {
            if (RT_DETECT_ABLATE != 8) a.hot_total[s] = (int32_t)hot_sum;
            a.large_any[s] = 0u;
            if (a.work_count && blockIdx.x == 0) *a.work_count = 0;  // (detect_group's work list: empty for the slot's next call)
        }
        const unsigned long long v = atomicAdd(&a.counters[0], (unsigned long long)n | (1ull << kTicketShift));
        const unsigned long long mask = (1ull << kTicketShift) - 1ull;
        const long long base = (long long)(v & mask);
        // a pool too short for this stream: its first records (in (bin, start) order) that still fit are delivered; the
        // call's total below tells the host how large a pool the call wants (rt_fetch grows it and runs the call again)
        long long fit = a.pool_cap - base;
        fit = fit < 0 ? 0 : (fit > n ? n : fit);
        sh_base = fit > 0 ? base : -1;
        sh_fit = (int)fit;
        if (RT_DETECT_ABLATE != 8) {
            a.rec_offset[s] = fit > 0 ? (int)base : 0;
            a.rec_count[s] = (int)fit;
        }
        if ((v >> kTicketShift) + 1ull == (unsigned long long)gridDim.x) {
            // every workgroup has added its records: publish the counter words, leave them zero for the slot's next call
            const unsigned long long total = __hip_atomic_load(&a.counters[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & mask;
            unsigned long long flags = __hip_atomic_load(&a.counters[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (total > (unsigned long long)a.pool_cap) flags |= kFlagRecOverflow;
            a.host_counters[0] = total;
            a.host_counters[1] = __hip_atomic_load(&a.counters[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a.host_counters[2] = flags;
            a.host_counters[3] = (unsigned long long)gridDim.x;
            a.host_counters[4] = __hip_atomic_load(&a.counters[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.seg_total) *a.host_seg_total = __hip_atomic_load(a.seg_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int i = 0; i < kCounterWords; ++i) __hip_atomic_store(&a.counters[i], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    const long long base = sh_base;
    const int n_fit = sh_fit;
    if (n == 0 || base < 0) return;
    const float cal_db = a.cal_s ? a.cal_s[s] : a.dp.cal_db;
    const rt_record *raw = a.raw + (int64_t)s * a.rec_cap;
    for (int i0 = 0; i0 < n; i0 += kFinalTile) {  // the records this pass ranks (one per thread)
        const int i = i0 + tid;
        rt_record mine{};
        long long ts_i = 0, dur_i = 0;
        float mx_i = 0.f;
        if (i < n) {
            mine = raw[i];
            ts_i = timedelta_us(start_time(a.dp, mine.start));
            dur_i = timedelta_us(run_duration(a.dp, mine.start, mine.end));
            mx_i = db10(mine.max_p) - cal_db;
        }
        int rank = 0, shadow = 0;
        for (int j0 = 0; j0 < n; j0 += kFinalTile) {  // ... against every record, a tile at a time
            __syncthreads();
            if (j0 == i0) {
                t_rec[tid] = mine;
                t_rec[tid].reserved = __float_as_int(mx_i);
                t_ts[tid] = ts_i;
                t_dur[tid] = dur_i;
            } else if (j0 + tid < n) {
                rt_record r = raw[j0 + tid];
                r.reserved = __float_as_int(db10(r.max_p) - cal_db);
                t_rec[tid] = r;
                t_ts[tid] = timedelta_us(start_time(a.dp, r.start));
                t_dur[tid] = timedelta_us(run_duration(a.dp, r.start, r.end));
            }
            __syncthreads();
            const int nj = (n - j0 < kFinalTile) ? (n - j0) : kFinalTile;
            if (i < n) {
                for (int j = 0; j < nj; ++j) {
                    const rt_record &rj = t_rec[j];
                    if (rj.fi < mine.fi || (rj.fi == mine.fi && rj.start < mine.start)) ++rank;
                    if (shadowed_by(ts_i, dur_i, mx_i, t_ts[j], t_dur[j], __int_as_float(rj.reserved))) shadow = 1;
                }
            }
        }
        if (i < n) {
            mine.shadowed = shadow;
            mine.reserved = 0;
            if (rank < n_fit) (RT_DETECT_ABLATE == 9 ? a.raw[(int64_t)s * a.rec_cap + rank] : a.records[base + rank]) = mine;
        }
    }
}

// Device form of rt::scan_dense_row (same decisions): the row is read in blocks of 16 time
// steps so the 16 independent loads overlap -- one dependent load per step made the kernel
// latency-bound at ~1 us per cell.
template <class OnRun>
__device__ __forceinline__ bool scan_dense_row_blocked(const DetectParams &p, const float *row, int64_t stride,
                                                       double row_sum, int t_begin, int t_end, float *avg_out,
                                                       OnRun on_run) {
    const int T = p.n_seg;
    constexpr int B = 16;
    if (row_sum < 0.0) {  // row mean not known (caller-supplied spectrogram): sum it here, in t order
        double sum = 0.0;
        for (int t0 = 0; t0 < T; t0 += B) {
            float v[B];
#pragma unroll
            for (int k = 0; k < B; ++k) v[k] = (t0 + k < T) ? row[(int64_t)(t0 + k) * stride] : 0.f;
#pragma unroll
            for (int k = 0; k < B; ++k)
                if (t0 + k < T) sum += (double)v[k];
        }
        row_sum = sum;
    }
    const float avg = (float)row_sum / (float)T;  // np.mean(row) (analyze.py:375)
    *avg_out = avg;
    // This thread owns the runs that START in [t_begin, t_end): a run already open at t_begin
    // belongs to the thread of the earlier range (which follows it past its own end).
    bool in_cont = (t_begin > 0) && cell_above(row[(int64_t)(t_begin - 1) * stride], avg, p.thr, p.snr);
    int b = -1;
    // The predicate is evaluated for a block of 16 cells without branches into a bit mask; the run logic then
    // only visits the cells where the mask changes (a cell-by-cell state machine cost ~100 instructions per
    // cell and wave, mostly scalar control flow: SQ_ACTIVE_INST_ANY at the issue limit, detect_dense 1.4 ms).
    uint32_t prev = in_cont ? 1u : 0u;  // was the cell before this block above?
    for (int t0 = t_begin; t0 < T; t0 += B) {
        float v[B];
#pragma unroll
        for (int k = 0; k < B; ++k) v[k] = (t0 + k < T) ? row[(int64_t)(t0 + k) * stride] : 0.f;
        const int nv = (T - t0 < B) ? (T - t0) : B;  // valid cells of the block
        uint32_t m = 0;
#pragma unroll
        for (int k = 0; k < B; ++k) {
            const bool ab = !(v[k] < p.thr) & !(v[k] / avg < p.snr);  // == cell_above(), without the early exits
            m |= (ab ? 1u : 0u) << k;
        }
        const uint32_t valid = (1u << nv) - 1u;
        m &= valid;
        uint32_t tr = (m ^ ((m << 1) | prev)) & valid;  // bit k: cell k differs from the cell before it
        prev = (m >> (nv - 1)) & 1u;
        const int kc = t_end - 1 - t0;  // the last cell of the own range (may lie in an earlier or a later block)
        int pos = 0;
        bool done = false;
        for (;;) {
            const int kt = tr ? __builtin_ctz(tr) : nv;  // next change of state, or the end of the block
            // cells pos..kt-1 keep the state: with nothing open, the first of them at or past the end of the own
            // range ends the scan (`t + 1 >= t_end && b < 0` of the cell-by-cell form)
            const int chk = pos > kc ? pos : kc;
            if (b < 0 && chk < kt) {
                done = true;
                break;
            }
            if (kt >= nv) break;
            const int t = t0 + kt;
            if ((m >> kt) & 1u) {
                if (t < t_end) b = t;  // a run starts in the own range
            } else if (in_cont) {
                in_cont = false;  // the run that was open at t_begin (an earlier range's) ends
            } else if (b >= 0) {
                const int rb = b;
                b = -1;
                on_run(rb, t, avg);
            }
            tr &= tr - 1u;
            pos = kt + 1;
            if (kt >= kc && b < 0) {
                done = true;
                break;
            }
        }
        if (done) break;
    }
    // a run still open here touches the end of the buffer: skipped (analyze.py:415)
    return true;
}

// One workgroup per stream.  Phase 1: one thread per bin (strided) scans its
// row sequentially in time -- the reference's row scan (analyze.py:357-450) in
// run-based form.  Phase 2/3 as in detect_sparse.
// GLOBAL (a handle whose record capacity has grown beyond kDenseLdsRecords): the unordered list lives in the stream's raw-record
// area in global memory, like the sparse path's, and finalize_records -- launched behind this kernel -- orders it, applies the
// shadow verdicts and closes the call.
template <bool GLOBAL>
__global__ __launch_bounds__(kDetBlock) void detect_dense(const DetectArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int s_pos = blockIdx.x;
    const int s = a.stream_list ? a.stream_list[s_pos] : s_pos;
    const int tid = threadIdx.x;
    const int F = a.n_bins;
    const int T = a.dp.n_seg;
    unsigned char *ptr = smem;
    RecLds l;
    if constexpr (GLOBAL) {
        l.rec = a.raw + (int64_t)s * a.rec_cap;
        l.ts_us = nullptr;
        l.dur_us = nullptr;
        l.count = a.raw_count + s;  // (zero between calls: finalize_records leaves it so)
    } else {
        l = carve_rec_lds(ptr, a.rec_cap);
        if (tid == 0) *l.count = 0;
    }
    __syncthreads();

    const DetectParams dp = stream_params(a, s);
    const float *sp = a.spec + (int64_t)s_pos * T * F;
    // work items = (bin, time range): with few bins the time axis is split so that all 1024
    // threads scan (runs are owned by the range they start in)
    int Q = 1;
    if (a.psum)
        while (F * Q * 2 <= kDetBlock && Q * 2 * 64 <= T) Q *= 2;
    const int span = (T + Q - 1) / Q;
    for (int item = tid; item < F * Q; item += kDetBlock) {
        const int fi = item % F, q = item / F;
        const int t_begin = q * span;
        const int t_end = (t_begin + span < T) ? t_begin + span : T;
        if (t_begin >= T) continue;
        const float *row = sp + fi;
        PrevCells prev{a.prev + ((int64_t)s * a.prev_cols + a.prev_cols) * F + fi, F};
        double row_sum = -1.0;  // same partial sums (and bits) as the sparse path
        if (a.psum) row_sum = row_sum_from_partials(a.psum + (int64_t)s * a.chunks * F + fi, a.chunks, F);
        float av = 0.f;
        auto on_run = [&](int b, int e, float avg) {
            int start;
            if (gate_run(dp, b, e, avg, prev, &start)) push_candidate(a, l, s, fi, start, e, avg, 0);
        };
        scan_dense_row_blocked(dp, row, F, row_sum, t_begin, t_end, &av, on_run);
    }
    int nrec;
    if constexpr (GLOBAL) {
        __threadfence();
        __syncthreads();
        nrec = __hip_atomic_load(l.count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (finalize_records clamps, flags and reports what was wanted)
        if (nrec > a.rec_cap) nrec = a.rec_cap;
    } else {
        nrec = settled_count(a, l);
    }

    for (int c = tid >> 6; c < nrec; c += kDetBlock / 64) {
        rt_record &r = l.rec[c];
        const int start = r.start;
        const float *row = sp + r.fi;
        PrevCells prev{a.prev + ((int64_t)s * a.prev_cols + a.prev_cols) * F + r.fi, F};
        auto cell = [&](int k) -> float {
            const int t = start + k;
            return t < 0 ? prev(-t) : row[(int64_t)t * F];
        };
        const RunStats st = run_stats_wave(r.end - start, cell);
        if ((tid & 63) == 0) {
            r.max_p = st.max_p;
            r.mean_p = st.mean_p;
            r.std_db = st.std_db;
        }
    }
    if constexpr (!GLOBAL) {
        __syncthreads();
        publish_records(a, l, s, nrec);
        if (tid == 0) close_call(a);
    }
}

}  // namespace rt

#include "rt_scan64.h"  // the nperseg-4096 scan (one wave per segment)
#include "rt_scan_wg.h"  // the nperseg-8192 / 16 384 scan (one workgroup per segment)
#endif
