// rt_scan_wg.h -- the nperseg-8192 / 16 384 scan: ONE WORKGROUP PER SEGMENT (included by rt_kernels.h, round 6).
//
//   stft_wg<BLK, MODE, U8>   N = 32 BLK points held by BLK = 256 / 512 threads, 32 per thread: the contract of stft_scan
//                            (rt_kernels.h) for the sparse path (MODE 0), the dense path (MODE 1) and the spectrogram alone (MODE 2):
//                            scipy.signal.spectrogram at radiotracking/analyze.py:234-241 for fft_nperseg = 8192 / 16 384, fused
//                            with what extract_signals (analyze.py:330-452) needs of it -- power, per-bin row sums, the look-back
//                            tail and either the candidate cells or the dense map.
//
// Round 5 served these sizes on the dense path only: stft_big (radix-2 in LDS: thirteen stages, seven LDS round trips of the whole
// segment, 6.3 ms per 13 GB = 2.1 TB/s) -> a 4-byte map per sample -> row_sums_dense -> detect_dense, 20 bytes of traffic per sample,
// 192 k MS/s at nperseg 8192.  Here the transform is three radix passes in REGISTERS with two LDS exchanges, and nothing but the
// candidate cells (or, on the dense path, the map once) leaves the kernel:
//     n = t + BLK j          thread t holds x[t + BLK j], j = 0 .. 31            (a piece j is BLK consecutive samples: whole lines)
//     pass A (in-thread)     A[t][k1] = sum_j w (x - mean) W32^(j k1),   times W_N^(t k1)   (the factor built from W^t, W^2t, .. W^16t)
//     exchange 1 (LDS)       row k1, column t                                     (32 rows of BLK + 2 complex values)
//     pass B (in-thread)     t = d + R c (R = BLK / 16):  D[d][p][k1] = sum_c A'[d + R c][k1] W16^(c p),  times W_BLK^(d p)
//     exchange 2 (LDS)       IN PLACE: the sixteen values of a (k1, d) pair go back to the places they came from, p for c -- no
//                            barrier between the reads of exchange 1 and these writes, and a (k1, p) pair's R values lie side by side
//     pass C (in-thread)     X[k1 + 32 (p + 16 q)] = sum_d D'[d][p][k1] W_R^(d q)           (R = 16: two pairs per thread; 32: one)
// and with pair u = k1 + 32 p taken by thread u mod BLK the result register r of thread t holds bin t + BLK r: every store of a
// spectrogram row or tail column is BLK consecutive floats, a thread's threshold bits are one 32-bit word.
// Bank conflicts (tools/lds_banks.py rules; rows of BLK + 2): all four accesses of a step conflict-free at BLK = 512, the
// ds_read_b64 of exchange 1 two-way at BLK = 256 -- ~580 LDS cycles per wave and segment, a third of what HBM takes to deliver it.
//
// Three workgroup barriers per step (behind the mean's partial sums -- which also frees the rows --, behind each exchange's
// writes).  The mean is subtracted first, in SciPy's order (_spectral_py.py:2191-2194), from the waves' float32 sums added in float64:
// no detrend-by-linearity form here, so no guard and no second launch.  The window comes from L2 in thread order (eight 16-byte
// loads per step, issued ahead of the next segment's samples: vector-memory operations return in order); the next segment's
// thirty-two samples are prefetched into registers.  Two workgroups per CU at BLK = 256 (66 KiB of rows each), one at 512: two
// waves per SIMD either way, up to 256 VGPRs.
//
// Work items: one chunk (segs_per_chunk segments) of one stream per workgroup, latest chunks first, walked downwards in time like
// every scan, so "the next cell is a candidate" is the previous step's word; the step below the chunk (segment c0 - 1) is taken only
// where a lowest cell of the chunk is hot (stft_scan: BELOW).  Row sums stay in registers over the chunk: one partial row per item,
// no reduction.  The look-back tail's columns are written whole (seventeen of 390 segments at 3.2 MS/s).
#ifndef RT_SCAN_WG_H
#define RT_SCAN_WG_H

namespace rt {

constexpr int kWgStage = 128;  // candidate cells staged per wave before a flush (1 KiB)

// diagnostic builds only (tools/variant.sh <name> -DRT_WG_ABL=mask; timing only, WRONG results): 1 = no window loads, 2 = no mean (and not
// its barrier), 4 = every load hits the same 64 segments of stream 0 (L2-resident: the kernel without HBM), 8 = no threshold test / emission,
// 16 = no row sums, no tail.  0 = the product.
#ifndef RT_WG_ABL
#define RT_WG_ABL 0
#endif
// Experiment, diagnostic builds only (-DRT_EXP_WG_HALF=1; measured and left off: EXPERIMENTS.md round 6, entry 24): nperseg 8192 by
// HALVES of the exchange -- rows k1 < 16 go through LDS first (pass B's pair e = 0 and, with pass C's pairs dealt (k1 mod 16, p) per
// half, pass C need only those), then rows k1 >= 16 through the same 33 KiB: three workgroups per CU instead of two, the next
// segment's samples requested into the registers each half's pass-A values leave.  Correct (the 8192 parity tests pass on it), but
// three workgroups per CU are 168 registers per thread, and the row sums (32), the twiddle bases (10), the prefetched samples (64),
// the other half's pass-A values (32) and a pair's sixteen (32) are 170 before any temporary: 154 registers in scratch, 12.3 against
// 4.5 ms per 13 GB.
#ifndef RT_EXP_WG_HALF
#define RT_EXP_WG_HALF 0
#endif
__host__ __device__ constexpr bool wg_half(int blk) { return RT_EXP_WG_HALF != 0 && blk == 256; }
__host__ __device__ constexpr int wg_row(int blk) { return blk + 2; }
__host__ __device__ constexpr size_t wg_lds_bytes(int blk) { return sizeof(cf) * (wg_half(blk) ? 16 : 32) * (size_t)wg_row(blk); }  // the exchange rows (dynamic LDS)
__host__ __device__ constexpr int wg_block(int nperseg) { return nperseg / 32; }                              // 256 / 512 threads

// v[k1] *= W_N^(t k1), k1 = 1 .. 31, from the five factors wb[i] = W_N^(t 2^i): a walk over the bits of k1, high to low, the
// product so far handed down -- 26 + 31 complex multiplications, five values live at a time (a table of the 31 factors per
// thread would be 62 registers, or as many bytes from L2 in every step as the samples take from HBM)
template <int BIT, int K, bool ONE>
__device__ __forceinline__ void wg_twiddle(cf (&v)[32], const cf (&wb)[5], cf f) {
    if constexpr (BIT < 0) {
        if constexpr (!ONE) v[K] = cmul(v[K], f);
    } else {
        wg_twiddle<BIT - 1, K, ONE>(v, wb, f);
        const cf g = ONE ? wb[BIT] : cmul(f, wb[BIT]);
        wg_twiddle<BIT - 1, K | (1 << BIT), false>(v, wb, g);
    }
}

// Candidate cells of one step: bit q of `emit` = the cell (seg, bin0 + bin_stride q) with power Pp[q] is kept.  Lane-centric staging as
// in stft_scan: a lane counts its cells, prefix and total from the bit planes of the counts, the powers picked out of the registers by a
// select tree; flushed with one returned atomic per bucket (flush_stage).  Wave-uniform control flow; rare.
template <int NP>
__device__ __forceinline__ void wg_emit(const StftParams &p, int s, int seg, const float (&Pp)[NP], uint32_t emit, int bin0, int bin_stride, uint2 *stg,
                                        int &stg_n, bool &gave_up) {
    if (gave_up || __builtin_amdgcn_ballot_w64(emit != 0) == 0) return;
    asm volatile("" : "+v"(bin0));  // (opaque: the key bases stay out of the step loop's registers)
    const int lane = threadIdx.x & 63;
    const int cnt = __builtin_popcount(emit);
    int off = 0, need = 0;
    constexpr int PLANES = (NP == 32) ? 6 : 5;
#pragma unroll
    for (int k = 0; k < PLANES; ++k) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(((cnt >> k) & 1) != 0);
        off += (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0)) << k;
        need += __builtin_popcountll(m) << k;
    }
    if (stg_n + need > kWgStage) {
        flush_stage(p, s, stg, stg_n);
        stg_n = 0;
    }
    if (need <= kWgStage) {
        uint32_t e = emit;
        int o = stg_n + off;
        while (__builtin_amdgcn_ballot_w64(e != 0u) != 0ull) {  // (wave-uniform)
            if (e) {
                const int q = __builtin_ctz(e);
                e &= e - 1u;
                const uint32_t key = ((uint32_t)(bin0 + bin_stride * q) << p.tbits) | (uint32_t)seg;
                stg[o++] = make_uint2(key, __float_as_uint(pick_range<0, NP>(Pp, q)));
            }
        }
        stg_n += need;
    } else {
        // more than a staging area in one step (dense input): straight to memory -- unless one of the stream's lists has overflowed
        // already: then the call is analysed again dense (AUTO) or fails (SPARSE) whatever else is emitted
        const uint32_t cnt16 = lane < kBuckets ? __hip_atomic_load(&p.hot_count[s * kBuckets + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        gave_up = __builtin_amdgcn_ballot_w64(cnt16 > (uint32_t)p.hot_cap) != 0;
#pragma unroll 1
        for (int q = 0; q < NP && !gave_up; ++q) {
            if (emit & (1u << q)) {
                const int bin = bin0 + bin_stride * q;
                const int bkt = bin & (kBuckets - 1);
                const uint32_t slot = atomicAdd(&p.hot_count[s * kBuckets + bkt], 1u);
                if (slot < (uint32_t)p.hot_cap)
                    p.hot[((int64_t)s * kBuckets + bkt) * p.hot_cap + slot] = make_uint2(((uint32_t)bin << p.tbits) | (uint32_t)seg, __float_as_uint(pick_range<0, NP>(Pp, q)));
            }
        }
    }
}

// WCOS: the window is a cosine sum of order <= 1 (hamming, hann, boxcar -- what get_window makes of them; the host checks the fit, as
// for the detrend by linearity): w[t + BLK j] = c0 + c1 cos(alpha_t + beta_j), alpha_t = 2 pi t / N (the thread's own W_N^t holds its
// cosine and sine), beta_j = 2 pi j / 32 (constants) -- two fused multiply-adds per sample instead of a table in registers: the table's
// 32 registers, live from the request to the multiplication, were what made the kernel spill.  Other windows keep the table (from L2).
template <int BLK, int MODE, bool U8, bool WCOS>
__global__ __launch_bounds__(BLK, wg_half(BLK) ? 3 : (BLK == 256 ? 2 : 1)) void stft_wg(const StftParams p) {
    using raw_t = typename std::conditional<U8, iq_u8, cf>::type;
    static_assert(BLK == 256 || BLK == 512, "nperseg 8192 / 16384");
    static_assert(MODE == 0 || MODE == 1 || MODE == 2, "sparse, dense, spectrogram only");
    constexpr int N = 32 * BLK, R = BLK / 16, S1 = wg_row(BLK), NW = BLK / 64;
    constexpr bool EMIT = (MODE == 0), SUMS = (MODE != 2), SPEC = (MODE == 1 || MODE == 2);
    constexpr bool HALF = wg_half(BLK);
    extern __shared__ __attribute__((aligned(16))) unsigned char wg_smem[];
    cf *const xs = reinterpret_cast<cf *>(wg_smem);  // [32][S1]
    __shared__ double red[2 * NW];
    __shared__ __attribute__((aligned(16))) cf tw2_lds[R * 16];  // W_BLK^(d p)
    __shared__ uint2 stage_lds[EMIT ? NW * kWgStage : 1];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = p.n_seg, L = p.segs_per_chunk;

    for (int idx = tid; idx < R * 16; idx += BLK) tw2_lds[idx] = p.tw2[idx];
    cf wb[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) wb[i] = p.tw1[i * BLK + tid];

    const int item = blockIdx.x;
    const int s_pos = item % p.n_streams;
    const int s = p.stream_list ? p.stream_list[s_pos] : s_pos;
    const int cb = p.chunks - 1 - item / p.n_streams;  // latest chunks first
    const int c0 = cb * L;
    const raw_t *stream_iq = reinterpret_cast<const raw_t *>(p.iq) + (int64_t)s * p.stream_stride;
    const float thr = p.thr_s ? p.thr_s[s] : p.thr;  // (uniform)

    float acc[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) acc[r] = 0.f;
    constexpr int PAIRS = (R == 16) ? 2 : 1, NP = 32 / PAIRS;  // (k1, p) pairs a thread finishes in pass C, bins of each
    uint32_t next_hot[PAIRS];  // hot bits of the segment one later in time, per pair (bit q: bin t + BLK (e + PAIRS q))
#pragma unroll
    for (int e = 0; e < PAIRS; ++e) next_hot[e] = 0u;
    uint2 *const stg = stage_lds + (EMIT ? wave * kWgStage : 0);
    int stg_n = 0;          // wave-uniform
    bool gave_up = false;   // wave-uniform: a list of this stream has overflowed

    raw_t nxt[32];
    // pieces j0 .. j0 + 15 of a segment (HALF: a half is requested where the registers of pass A's values of the same half are free)
    auto request_half = [&](int seg_req, int j0) {
        const int sg = __builtin_amdgcn_readfirstlane(seg_req);
        // (a segment outside the buffer gets an empty descriptor: its loads return zeros)
        const raw_t *base = (RT_WG_ABL & 4) ? reinterpret_cast<const raw_t *>(p.iq) + (int64_t)(sg & 63) * N : stream_iq + (int64_t)(sg < 0 ? 0 : sg) * N;
        const rsrc_t r = make_rsrc(base, (sg >= 0 && sg < T) ? (uint32_t)(N * sizeof(raw_t)) : 0u);
#pragma unroll
        for (int j = 0; j < 16; ++j) nxt[j0 + j] = buf_load_iq(r, tid * (int)sizeof(raw_t), BLK * (j0 + j) * (int)sizeof(raw_t), raw_t{});
    };
    auto request = [&](int seg_req) {
        request_half(seg_req, 0);
        request_half(seg_req, 16);
    };
    // the window in thread order, from L2: requested for the NEXT step once a step's last LDS reads are issued (its latency hides behind
    // the epilogue; at the head of the step it was exposed behind the samples': vector-memory operations return in order)
    float w[WCOS ? 1 : 32];
    const float wc0 = p.lin_c[0], wca = p.lin_c[1] * wb[0].x, wcb = p.lin_c[1] * wb[0].y;  // WCOS: c0, c1 cos(alpha), -c1 sin(alpha)
    auto request_window = [&]() {
        if constexpr (WCOS) {
        } else if constexpr (RT_WG_ABL & 1) {
#pragma unroll
            for (int j = 0; j < 32; ++j) w[j] = 0.5f + 0.01f * j;
        } else {
            const rsrc_t rw = make_rsrc(p.window_t, (uint32_t)(N * sizeof(float)));
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const buf_f4 w4 = raw_buffer_load_f4(rw, tid * 128, q * 16, 0);
                w[4 * q] = w4.x;  w[4 * q + 1] = w4.y;  w[4 * q + 2] = w4.z;  w[4 * q + 3] = w4.w;
            }
        }
    };
    int n_steps = L;
    request(c0 + L - 1);
    request_window();
    __syncthreads();  // the twiddle table is staged

    for (int i = 1; i <= n_steps; ++i) {
        const int seg = c0 + L - i;
        const bool halo = i > L;               // the step below the chunk: no sums, no tail, no map
        const bool active = seg < T && seg >= 0;  // (workgroup-uniform)
        cf v[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) v[j] = to_cf(nxt[j]);
        if (!active) {  // (the upper steps of a stream's last chunk; uniform: no barrier is skipped by part of a workgroup)
            if (i < n_steps) request(seg - 1);
            continue;
        }

        // detrend='constant' (scipy _signaltools.py:3926): the segment's mean -- a tree of float32 sums per thread, the wave's by DPP folds,
        // the waves' in float64, in order
        if constexpr (!(RT_WG_ABL & 2)) {
            cf s16[16], s8[8], s4[4];
#pragma unroll
            for (int j = 0; j < 16; ++j) s16[j] = cadd(v[j], v[j + 16]);
#pragma unroll
            for (int j = 0; j < 8; ++j) s8[j] = cadd(s16[j], s16[j + 8]);
#pragma unroll
            for (int j = 0; j < 4; ++j) s4[j] = cadd(s8[j], s8[j + 4]);
            const cf s1 = cadd(cadd(s4[0], s4[2]), cadd(s4[1], s4[3]));
            // (the wave's 64 partial sums by DPP folds in float32 -- wave_sum, as every scan up to nperseg 4096 adds a segment's samples --,
            // the waves' sums in float64: sixty instructions fewer per step than float64 butterflies over the lanes)
            const cf sw = wave_sum<64>(s1);
            if (lane == 0) {
                red[2 * wave] = (double)sw.x;
                red[2 * wave + 1] = (double)sw.y;
            }
        }
        if constexpr (!(RT_WG_ABL & 2)) __syncthreads();  // (also: every wave has read the rows of the step before)
        {
            double tx = 0.0, ty = 0.0;
            if constexpr (!(RT_WG_ABL & 2)) {
#pragma unroll
                for (int wv = 0; wv < NW; ++wv) {
                    tx += red[2 * wv];
                    ty += red[2 * wv + 1];
                }
            }
            const float mx = (float)(tx / (double)N), my = (float)(ty / (double)N);
            // (WCOS: the thread's two factors made opaque here -- the 32 coefficients are loop-invariant, and hoisted out of the step loop
            // they are the table in registers again: 29 of them went to scratch)
            float wca_ = wca, wcb_ = wcb;
            asm volatile("" : "+v"(wca_), "+v"(wcb_));
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                float wj;
                if constexpr (WCOS) {
                    // cos(alpha + beta_j) = cos a cos b - sin a sin b; W64^(2 j) = (cos b, -sin b) for j < 16, the negative of entry j - 16 beyond
                    const float cb = (j < 16) ? kW64Re[2 * (j & 15)] : -kW64Re[2 * (j & 15)], msb = (j < 16) ? kW64Im[2 * (j & 15)] : -kW64Im[2 * (j & 15)];
                    wj = __builtin_fmaf(wca_, cb, __builtin_fmaf(-wcb_, msb, wc0));
                } else {
                    wj = w[j];
                }
                v[j] = cf{(v[j].x - mx) * wj, (v[j].y - my) * wj};  // window (times sqrt(scale))
            }
        }
        // WCOS: the next segment's samples are requested here, a whole transform ahead of their use (without the window table there are
        // registers for them beside pass A); with the table, behind exchange 1 (below)
        if constexpr (WCOS && !HALF) {
            __builtin_amdgcn_sched_barrier(0);
            if (i < n_steps) request(seg - 1);
        }
        // pass A and its twiddles
        // (scheduling fences between the phases: left alone, hipcc hoists the LDS reads of both pairs of a pass and the twiddle table
        // above the pass before -- a schedule its register allocator then cannot hold: 60 - 100 registers in scratch)
        __builtin_amdgcn_sched_barrier(0);
        dft32(v);
        __builtin_amdgcn_sched_barrier(0);
        {
            // (opaque copies: the 26 products of the five factors are loop-invariant too -- hoisted, they are 52 registers for good)
            cf wq[5];
#pragma unroll
            for (int b = 0; b < 5; ++b) {
                wq[b] = wb[b];
                asm volatile("" : "+v"(wq[b].x), "+v"(wq[b].y));
            }
            wg_twiddle<4, 0, true>(v, wq, cf{1.f, 0.f});
        }
        __builtin_amdgcn_sched_barrier(0);
        const int tcol = seg - (T - p.tail_cols);
        const bool to_tail = SUMS && !(RT_WG_ABL & 16) && !halo && tcol >= 0;
        // (rows leave through buffer stores: one descriptor per row, ONE address register -- global stores 1 .. 2 KiB apart each want a
        // 64-bit address of their own)
        const rsrc_t rs_spec = make_rsrc(SPEC ? p.spec + ((int64_t)(p.spec_by_stream ? s : s_pos) * T + seg) * N : nullptr, SPEC ? (uint32_t)(N * sizeof(float)) : 0u);
        const rsrc_t rs_tail = make_rsrc(to_tail ? p.tail + ((int64_t)s * p.tail_cols + tcol) * N : nullptr, to_tail ? (uint32_t)(N * sizeof(float)) : 0u);
        uint32_t any_hot = 0u;
      if constexpr (HALF) {
        // The exchanges by halves: rows k1 = 16 h .. 16 h + 15 through 16 rows of LDS.  Pass B's pair of half h is (k1 = 16 h + tid / 16,
        // d = tid % 16) -- the whole-segment form's pair e = h --, pass C's is (k1 = 16 h + tid % 16, p = tid / 16): register q of the half
        // holds bin 16 h + tid % 16 + 32 (tid / 16) + 512 q, sixteen consecutive bins per sixteen lanes.
        const int bin_t = (tid & 15) + 32 * (tid >> 4);  // the thread's bin of half 0, q = 0
        const float4 *const t2 = reinterpret_cast<const float4 *>(tw2_lds + (tid % R) * 16);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            __builtin_amdgcn_sched_barrier(0);
            if (h == 1) __syncthreads();  // (pass C of half 0 has read its rows)
#pragma unroll
            for (int k = 0; k < 16; ++k) xs[k * S1 + tid] = v[16 * h + k];
            // the next segment's pieces 16 h .. 16 h + 15 into the registers this half's values have just left
            __builtin_amdgcn_sched_barrier(0);
            if (i < n_steps) request_half(seg - 1, 16 * h);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            {
                cf *const col = xs + (tid / R) * S1 + (tid % R);
                cf u[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) u[c] = col[R * c];
                dft16(u);
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const float4 t = t2[kk];
                    if (kk) u[2 * kk] = cmul(u[2 * kk], cf{t.x, t.y});
                    u[2 * kk + 1] = cmul(u[2 * kk + 1], cf{t.z, t.w});
                }
#pragma unroll
                for (int pp = 0; pp < 16; ++pp) col[R * pp] = u[pp];
            }
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            const float4 *row = reinterpret_cast<const float4 *>(xs + (tid & 15) * S1 + R * (tid >> 4));
            cf u[16];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 q4 = row[j];
                u[2 * j] = cf{q4.x, q4.y};
                u[2 * j + 1] = cf{q4.z, q4.w};
            }
            // (the step's last LDS reads are issued: the next step's window -- its latency hides behind this epilogue)
            if (h == 1 && i < n_steps) request_window();
            dft16(u);
            float Pp[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int r = 16 * h + q;  // the result register: bin 16 h + bin_t + 512 q
                Pp[q] = __builtin_fmaf(u[q].x, u[q].x, u[q].y * u[q].y);
                if constexpr (SUMS && !(RT_WG_ABL & 16)) {
                    if (!halo) acc[r] += Pp[q];
                }
                if constexpr (SPEC) raw_buffer_store_f1(Pp[q], rs_spec, bin_t * 4, (16 * h + 512 * q) * 4, 0);
                if constexpr (SUMS) {
                    if (to_tail) raw_buffer_store_f1(Pp[q], rs_tail, bin_t * 4, (16 * h + 512 * q) * 4, 0);
                }
            }
            if constexpr (EMIT && !(RT_WG_ABL & 8)) {
                float mxp = __builtin_fmaxf(Pp[0], Pp[1]);
#pragma unroll
                for (int q = 2; q < 16; ++q) mxp = __builtin_fmaxf(mxp, Pp[q]);
                uint32_t hot = 0;
                if (!(mxp < thr)) {
#pragma unroll
                    for (int q = 15; q >= 0; --q) hot = (hot << 1) | ((Pp[q] < thr) ? 0u : 1u);
                }
                const uint32_t emit = halo ? (next_hot[h] & ~hot) : (hot | next_hot[h]);
                wg_emit<16>(p, s, seg, Pp, emit, 16 * h + bin_t, 512, stg, stg_n, gave_up);
                next_hot[h] = hot;
                any_hot |= hot;
            }
        }
      } else {
        // exchange 1: row k1, column t
#pragma unroll
        for (int k1 = 0; k1 < 32; ++k1) xs[k1 * S1 + tid] = v[k1];
        // The next segment's samples are requested HERE, where the thread's 32 values have gone to LDS: requested at the head of the
        // step their 64 registers were live beside the samples', the window's and pass A's temporaries, and the kernel spilled (85
        // registers in scratch; 65 when requested behind the window).  They still have passes B and C and both exchanges to land.
        if constexpr (!WCOS) {
            __builtin_amdgcn_sched_barrier(0);
            if (i < n_steps) request(seg - 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        // pass B on the thread's two (k1, d) pairs, written back in place (p for c)
        {
            const float4 *const t2 = reinterpret_cast<const float4 *>(tw2_lds + (tid % R) * 16);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
#ifndef RT_EXP_WG_NOFENCE  // (diagnostic builds: A/B of the fences between the pairs)
                __builtin_amdgcn_sched_barrier(0);
#endif
                const int pi = tid + BLK * e;
                cf *const col = xs + (pi / R) * S1 + (pi % R);
                cf u[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) u[c] = col[R * c];
                dft16(u);
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const float4 t = t2[kk];
                    if (kk) u[2 * kk] = cmul(u[2 * kk], cf{t.x, t.y});
                    u[2 * kk + 1] = cmul(u[2 * kk + 1], cf{t.z, t.w});
                }
#pragma unroll
                for (int pp = 0; pp < 16; ++pp) col[R * pp] = u[pp];
            }
        }
        __syncthreads();
        // pass C, a (k1, p) pair at a time: its R values lie side by side; |X|^2 (scipy _spectral_py.py:2126-2128); the pair's powers are
        // consumed as they come -- row sums, map, tail column, threshold test and emission -- so that no more than one pair's are live
#pragma unroll
        for (int e = 0; e < PAIRS; ++e) {
#ifndef RT_EXP_WG_NOFENCE
            __builtin_amdgcn_sched_barrier(0);
#endif
            const int ui = tid + BLK * e;
            const float4 *row = reinterpret_cast<const float4 *>(xs + (ui % 32) * S1 + R * (ui / 32));
            cf u[R];
#pragma unroll
            for (int j = 0; j < R / 2; ++j) {
                const float4 q4 = row[j];
                u[2 * j] = cf{q4.x, q4.y};
                u[2 * j + 1] = cf{q4.z, q4.w};
            }
            // (the step's last LDS reads are issued: the next step's window -- its latency hides behind this epilogue)
            if (e == PAIRS - 1 && i < n_steps) request_window();
            if constexpr (R == 16) dft16(u); else dft32(u);
            float Pp[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const int r = e + PAIRS * q;  // the result register: bin tid + BLK r
                Pp[q] = __builtin_fmaf(u[q].x, u[q].x, u[q].y * u[q].y);
                if constexpr (SUMS && !(RT_WG_ABL & 16)) {
                    if (!halo) acc[r] += Pp[q];
                }
                if constexpr (SPEC) raw_buffer_store_f1(Pp[q], rs_spec, tid * 4, BLK * r * 4, 0);
                if constexpr (SUMS) {
                    if (to_tail) raw_buffer_store_f1(Pp[q], rs_tail, tid * 4, BLK * r * 4, 0);
                }
            }
            if constexpr (EMIT && !(RT_WG_ABL & 8)) {
                // candidates are rare: one maximum over the pair's cells, the per-cell tests only where it fires (a NaN cell means the
                // whole segment is NaN: the maximum is, and `!(m < thr)` holds as for the reference's `not (P < thr)`)
                float mxp = __builtin_fmaxf(Pp[0], Pp[1]);
#pragma unroll
                for (int q = 2; q < NP; ++q) mxp = __builtin_fmaxf(mxp, Pp[q]);
                uint32_t hot = 0;
                if (!(mxp < thr)) {
#pragma unroll
                    for (int q = NP - 1; q >= 0; --q) hot = (hot << 1) | ((Pp[q] < thr) ? 0u : 1u);
                }
                // a cell is kept if it is a candidate itself or directly precedes one (T11); below the chunk only the cells that precede
                // a hot one and are not hot themselves (those their owner emits)
                const uint32_t emit = halo ? (next_hot[e] & ~hot) : (hot | next_hot[e]);
                wg_emit<NP>(p, s, seg, Pp, emit, tid + BLK * e, BLK * PAIRS, stg, stg_n, gave_up);
                next_hot[e] = hot;
                any_hot |= hot;
            }
        }
      }
        // (the rows are rewritten behind the next step's first barrier)
        if constexpr (EMIT && !(RT_WG_ABL & 8)) {
            if (i == L) {
                // a lowest cell of the chunk is a candidate: the cell before it belongs to the chunk below, whose owner cannot know
                if (__syncthreads_or(c0 > 0 && any_hot != 0u)) {
                    n_steps = L + 1;
                    request(c0 - 1);
                    request_window();
                }
            }
        }
    }
    if constexpr (EMIT) {
        if (stg_n) flush_stage(p, s, stg, stg_n);
    }
    if constexpr (SUMS) {
        const rsrc_t rp = make_rsrc(p.psum + ((int64_t)s * p.blocks_per_stream + cb) * N, (uint32_t)(N * sizeof(float)));
        if constexpr (HALF) {
            const int bin_t = (tid & 15) + 32 * (tid >> 4);
#pragma unroll
            for (int r = 0; r < 32; ++r) raw_buffer_store_f1(acc[r], rp, bin_t * 4, (16 * (r / 16) + 512 * (r % 16)) * 4, 0);
        } else {
#pragma unroll
            for (int r = 0; r < 32; ++r) raw_buffer_store_f1(acc[r], rp, tid * 4, BLK * r * 4, 0);
        }
    }
}

}  // namespace rt
#endif
