// rt_scan64.h -- the nperseg-4096 scan: ONE WAVE PER SEGMENT (included by rt_kernels.h).
//
//   stft_scan64<MODE, U8, LIN>   the same contract, modes and outputs as stft_scan<16, MODE, U8, LIN> (rt_kernels.h), which it
//                                replaces: scipy.signal.spectrogram at radiotracking/analyze.py:234-241 for fft_nperseg = 4096,
//                                fused with what extract_signals (analyze.py:330-452) needs of it.
//
// Why a kernel of its own.  With 16 points per lane a 4096-point segment spans four waves: two LDS exchanges and two workgroup
// barriers per step, at which the waves of a group arrive ~1 000 cycles apart, a window table that does not fit LDS at three
// workgroups per CU, factored twiddles rebuilt every step (4.45 TB/s on the config-5 share, profiles/r03_k_*).  Here a segment
// is N = 64 x 64 points held by ONE wave, 64 points per lane:
//     n = l + 64 m        lane l holds x[l + 64 m], m = 0 .. 63        (every load instruction reads 512 consecutive bytes)
//     pass 1 (in-lane)    A[l][ka] = sum_m w x[l + 64 m] W64^(m ka)
//     exchange            lane ka receives A[n1][ka], n1 = 0 .. 63      (wave-private 64 x 64 transpose in LDS, real parts then
//                                                                        imaginary parts through the same rows; no barrier)
//     twiddle             x W_N^(ka n1),  n1 = c + 8 d:  W^(8 ka d) x W^(ka c), fourteen table rows in LDS
//     pass 2 (in-lane)    X[ka + 64 kb] = sum_n1 ... W64^(n1 kb)  ->  register kb of lane ka holds bin ka + 64 kb
// so bins are in natural order across lanes (every store of a spectrogram row or tail column is 256 consecutive bytes, no
// LDS staging), a lane's candidate bucket is lane & 15, and nothing a wave does depends on another wave: the only workgroup
// barrier of the kernel is the one behind the table staging.  No register prefetch (64 points per lane + 64 row sums leave no
// room): two waves per SIMD cover each other's HBM round trips, and the 64 loads of a segment are issued in the order pass 1
// consumes them, so its first 16-point transforms run while the rest of the segment is still arriving.
//
// One workgroup of eight waves per CU: 8 exchange areas of 64 rows x 68 floats (the pad columns of a wave's rows are its
// candidate staging area: 128 cells) + the window table (16 KiB, in the order the lanes read it) + 16 twiddle rows
// = 163 840 B, all of a CU's LDS.
//
// Work items are per WAVE: an item is one chunk (segs_per_chunk segments) of one stream, latest chunks first; every wave of
// the chip-filling grid draws further items from StftParams::work until none is left (all modes; a selective pass skips the
// items its plan left empty).  Per-lane bit words (chunk bits, threshold bits, cells to emit) are 64-bit: bit kb of lane l is
// bin l + 64 kb -- a row of them is the same 512 bytes the 16-bit words of 256 lanes were, and the planning kernels
// (plan_pass_b, plan_runs) are bitwise, so they serve both layouts unchanged.
#ifndef RT_SCAN64_H
#define RT_SCAN64_H

namespace rt {

constexpr int kW64Waves = 8;                          // waves per workgroup: two per SIMD, one workgroup per CU
constexpr int kW64Block = 64 * kW64Waves;
constexpr int kW64Row = 68;                           // exchange row stride in floats: 16-byte aligned rows, conflict-free columns
constexpr int kW64XchFloats = 64 * kW64Row;           // one wave's exchange area
constexpr int kW64Stage = 128;                        // candidate cells staged per wave (the pad columns: 64 rows x 16 B)
constexpr int kW64TwRows = 16;                        // rows 0..6: W_N^(8 l d), d = 1..7; rows 7..13: W_N^(l c), c = 1..7
constexpr size_t kW64LdsBytes = sizeof(float) * ((size_t)kW64Waves * kW64XchFloats + 2 * kW64TwRows * 64 + 4096);
static_assert(kW64LdsBytes == 163840, "the scan's LDS block is exactly a CU's 160 KiB");

// diagnostic builds only (tools/variant.sh <name> -DRT_W64_ABL=mask): 1 = no threshold test / emission, 2 = no tail columns,
// 4 = no detrend sum, 8 = flushes without their atomic, 16 = no step below the chunk, 32 = threshold bits but no emission
// -- wrong results, timing only.  0 = the product.
#ifndef RT_W64_ABL
#define RT_W64_ABL 0
#endif

__device__ void raw_buffer_store_f1(float v, rsrc_t rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.f32");

// OR over the 64 lanes of a wave, in a scalar register: DPP within the rows of 16, then the four row results
__device__ __forceinline__ uint32_t wave_or(uint32_t x) {
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xF, 0xF, true);  // row_half_mirror
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xF, 0xF, true);  // row_mirror
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 0) | (uint32_t)__builtin_amdgcn_readlane((int)x, 16) |
           (uint32_t)__builtin_amdgcn_readlane((int)x, 32) | (uint32_t)__builtin_amdgcn_readlane((int)x, 48);
}

// staged candidate cell i of a wave: the pad columns (64 .. 67) of its exchange rows, two cells per row
__device__ __forceinline__ uint2 *w64_stage_cell(float *rows, int i) {
    return reinterpret_cast<uint2 *>(rows + (i >> 1) * kW64Row + 64 + ((i & 1) << 1));
}

// flush_stage (rt_kernels.h) for cells staged in the pad columns
__device__ __forceinline__ void flush_stage64(const StftParams &p, int s, float *rows, int n) {
    const int lane = threadIdx.x & 63;
    wave_sync();
    uint32_t my_cnt = 0;  // lane b (< 16): cells of bucket b
#pragma nounroll
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const int bk = (i < n) ? (int)((w64_stage_cell(rows, i)->x >> p.tbits) & (kBuckets - 1)) : kBuckets;
#pragma nounroll
        for (int b = 0; b < kBuckets; ++b) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(bk == b);
            if (lane == b) my_cnt += (uint32_t)__builtin_popcountll(m);
        }
    }
    uint32_t slot_base = 0;
    if (lane < kBuckets && my_cnt && !(RT_W64_ABL & 8)) slot_base = atomicAdd(&p.hot_count[s * kBuckets + lane], my_cnt);
#pragma nounroll
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const uint2 e = (i < n) ? *w64_stage_cell(rows, i) : make_uint2(0u, 0u);
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
    wave_sync();
}

// a lane's 64-bit word of per-bin bits as two 32-bit halves (bit kb of the word = bin lane + 64 kb)
struct bits64 {
    uint32_t lo, hi;
};
__device__ __forceinline__ bits64 operator|(bits64 a, bits64 b) { return bits64{a.lo | b.lo, a.hi | b.hi}; }
__device__ __forceinline__ bits64 operator&(bits64 a, bits64 b) { return bits64{a.lo & b.lo, a.hi & b.hi}; }
__device__ __forceinline__ bits64 operator~(bits64 a) { return bits64{~a.lo, ~a.hi}; }
__device__ __forceinline__ bool any(bits64 a) { return (a.lo | a.hi) != 0u; }
__device__ __forceinline__ bits64 load_bits(const uint16_t *base, int64_t word) {
    const uint2 v = reinterpret_cast<const uint2 *>(base)[word];
    return bits64{v.x, v.y};
}
__device__ __forceinline__ void store_bits(uint16_t *base, int64_t word, bits64 b) {
    reinterpret_cast<uint2 *>(base)[word] = make_uint2(b.lo, b.hi);
}
// bit r of a word; r is a constant once the loops around it are unrolled
__device__ __forceinline__ bool bit_of(bits64 a, int r) { return ((r < 32 ? a.lo >> r : a.hi >> (r - 32)) & 1u) != 0u; }
__device__ __forceinline__ bool bit_of(uint32_t lo, uint32_t hi, int r) { return ((r < 32 ? lo >> r : hi >> (r - 32)) & 1u) != 0u; }

template <int MODE, bool U8, bool LIN>
__global__ __launch_bounds__(kW64Block, 1) void stft_scan64(const StftParams p) {
    using raw_t = typename std::conditional<U8, iq_u8, cf>::type;
    constexpr int N = 4096;
    constexpr bool EMIT = (MODE == 0 || MODE == 5 || MODE == 7);   // candidate cells go to the bucket lists
    constexpr bool FLAGS = (MODE == 0 || MODE == 4 || MODE == 6);  // threshold bits are kept
    constexpr bool SUMS = (MODE != 2 && MODE != 3 && MODE != 5 && MODE != 7);  // row sums and look-back tail
    constexpr bool LISTED = (MODE == 7);                           // the steps take the segments plan_runs listed
    constexpr bool TEST = (EMIT || FLAGS) && !(RT_W64_ABL & 1);       // the step ends with the threshold test

    __shared__ __attribute__((aligned(16))) float lds[kW64LdsBytes / sizeof(float)];
    cf *const tw = reinterpret_cast<cf *>(lds + kW64Waves * kW64XchFloats);                              // [16][64]
    float4 *const win = reinterpret_cast<float4 *>(lds + kW64Waves * kW64XchFloats + 2 * kW64TwRows * 64);  // [n0][j / 4][lane]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < kW64TwRows * 64; i += kW64Block) tw[i] = p.tw1[i];
    for (int i = tid; i < 1024; i += kW64Block) win[i] = reinterpret_cast<const float4 *>(p.window_t)[i];
    __syncthreads();  // the kernel's only workgroup barrier
    float *const rows = lds + wave * kW64XchFloats;

    const int T = p.n_seg, L = p.segs_per_chunk;
    const int n_items = p.n_streams * p.blocks_per_stream;
    const int n_waves = (int)gridDim.x * kW64Waves;
    int item = __builtin_amdgcn_readfirstlane((int)blockIdx.x * kW64Waves + wave);

    while (item < n_items) {
        uint32_t ticket = 0;
        if (lane == 0) ticket = atomicAdd(p.work, 1u);  // the next item: consumed at the end of this one, its latency is covered
        do {  // one item (break = nothing (more) to do for it)
            const int s_pos = item % p.n_streams;
            const int s = p.stream_list ? p.stream_list[s_pos] : s_pos;
            const int pb = item / p.n_streams;
            int chunk = p.blocks_per_stream - 1 - pb;  // latest chunks first: the ones that also write the tail run longest
            if constexpr (MODE == 5) {
                if (pb >= p.item_count[s]) break;
                chunk = p.item_chunks[(int64_t)s * p.blocks_per_stream + pb];
                if (chunk >= p.chunks) break;
            }
            int e0 = 0, n_mine = 0;  // MODE 7: this wave's entries of the stream's segment list
            if constexpr (LISTED) {
                const int cnt = p.seg_count[s];
                e0 = pb * L;
                if (e0 >= cnt) break;
                n_mine = cnt - e0 < L ? cnt - e0 : L;
            }
            const int c0 = chunk * L;
            const float thr = p.thr_s ? p.thr_s[s] : p.thr;  // wave-uniform
            const raw_t *stream_iq = reinterpret_cast<const raw_t *>(p.iq) + (int64_t)s * p.stream_stride;

            float acc[64];
#pragma unroll
            for (int r = 0; r < 64; ++r) acc[r] = 0.f;
            bits64 next_hot{0u, 0u};            // threshold bits of the segment one later in time
            bits64 allhot{~0u, ~0u};            // FLAGS: the chunk's bits so far
            bits64 need{~0u, ~0u};              // MODE 5: the lane's bins that may emit in every segment of the chunk
            uint32_t n_abs = 0;                 // MODE 4 / 6: this lane's cells at or above the absolute threshold
            int stg_n = 0;                      // wave-uniform fill level of the staging area
            bool gave_up = false;               // wave-uniform: a candidate list of this stream has overflowed
            if constexpr (MODE == 5) {
                const int64_t w = ((int64_t)s * p.chunks + chunk) * 64 + lane;
                need = load_bits(p.full, w);
                if (chunk > 0) need = need | load_bits(p.full, w - 64);
                if (chunk + 1 < p.chunks) need = need | load_bits(p.full, w + 64);
                bits64 need_run = need;
                if (chunk == 0) need_run = need_run | load_bits(p.first, (int64_t)s * L * 64 + lane);  // any run through t = 0
                if (__builtin_amdgcn_ballot_w64(any(need_run)) == 0ull) break;
            }
            if constexpr (LISTED) need = bits64{0u, 0u};  // (what a step emits comes with its segment)

            // Steps walk the chunk down from its latest segment.  A cell is a candidate cell if it passes the threshold or
            // directly precedes one that does (T11); for the chunk's lowest segment that concerns a cell of the chunk
            // below, whose owner cannot know: where (and only where) a lowest cell is hot the wave takes one more step on
            // segment c0 - 1 and emits the cells there that precede a hot one and are not hot themselves.
            int k7 = 0;
            int seg = LISTED ? p.seg_list[(int64_t)s * T + e0] : ((c0 + L < T ? c0 + L : T) - 1);
            bool halo = false;
            for (;;) {
                bits64 first_nxt{0u, 0u};  // MODE 5, chunk 0: the bins whose run through t = 0 reaches this segment; MODE 7: the cells to emit
                if constexpr (MODE == 5) {
                    if (chunk == 0 && seg < L) first_nxt = load_bits(p.first, ((int64_t)s * L + seg) * 64 + lane);
                }
                int seg_after = -1;
                if constexpr (LISTED) {
                    first_nxt = load_bits(p.cell_need, ((int64_t)s * T + seg) * 64 + lane);
                    if (k7 + 1 < n_mine) seg_after = p.seg_list[(int64_t)s * T + e0 + k7 + 1];
                }

                // ---- the segment's samples, in the order pass 1 consumes them: quarter n0 = elements m = n0 + 4 j
                cf v[64];
                cf sum{0.f, 0.f};  // of the raw samples, the same bits in every lane (detrend='constant', scipy _signaltools.py:3926)
                {
                    const rsrc_t r = make_rsrc(stream_iq + (int64_t)seg * N, (uint32_t)(N * sizeof(raw_t)));
                    raw_t raw[64];
#pragma unroll
                    for (int n0 = 0; n0 < 4; ++n0)
#pragma unroll
                        for (int j = 0; j < 16; ++j) {
                            const int m = n0 + 4 * j;
                            raw[m] = buf_load_iq(r, lane * (int)sizeof(raw_t), 64 * m * (int)sizeof(raw_t), raw_t{});
                        }
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (MODE == 3) {
                        // traffic calibration: the scan's exact load stream, nothing else
#pragma unroll
                        for (int m = 0; m < 64; ++m) {
                            const cf x = to_cf(raw[m]);
                            acc[0] += x.x + x.y;
                        }
                        if (seg > c0) { --seg; continue; }
                        break;
                    }
                    // the sum of a quarter's sixteen samples, one fixed order
                    auto quarter_sum = [&](int n0) {
                        cf a[4];
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            a[g] = cadd(cadd(to_cf(raw[n0 + 16 * g]), to_cf(raw[n0 + 16 * g + 4])), cadd(to_cf(raw[n0 + 16 * g + 8]), to_cf(raw[n0 + 16 * g + 12])));
                        return cadd(cadd(a[0], a[1]), cadd(a[2], a[3]));
                    };
                    cf mean{0.f, 0.f};
                    if constexpr (!LIN) {
                        // subtract-first form: the mean is needed before the window, i.e. once the whole segment has arrived
                        const cf q0 = quarter_sum(0), q1 = quarter_sum(1), q2 = quarter_sum(2), q3 = quarter_sum(3);
                        sum = wave_sum<64>(cadd(cadd(q0, q2), cadd(q1, q3)));
                        mean = cscale(sum, 1.0f / (float)N);
                    }
                    // ---- pass 1 (over m), a quarter at a time as it arrives: window, 16-point transform over j
                    cf q[4];
#pragma unroll
                    for (int n0 = 0; n0 < 4; ++n0) {
                        if constexpr (LIN) q[n0] = !(RT_W64_ABL & 4) ? quarter_sum(n0) : cf{1.f, 0.f};
                        cf a[16];
#pragma unroll
                        for (int jq = 0; jq < 4; ++jq) {
                            const float4 w4 = win[(n0 * 4 + jq) * 64 + lane];
                            const float w[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const cf x = to_cf(raw[n0 + 4 * (4 * jq + e)]);
                                a[4 * jq + e] = LIN ? cscale(x, w[e]) : cscale(csub(x, mean), w[e]);
                            }
                        }
                        dft16(a);
#pragma unroll
                        for (int j = 0; j < 16; ++j) v[n0 + 4 * j] = a[j];  // A[n0][k' = j]
                        __builtin_amdgcn_sched_barrier(0);  // (quarter by quarter: the window reads and transforms of later quarters stay behind)
                    }
                    // constant detrend by linearity: the sum is only needed after pass 2 (see below)
                    if constexpr (LIN) sum = wave_sum<64>(cadd(cadd(q[0], q[2]), cadd(q[1], q[3])));
                }
                dft64_finish(v);  // lane l now holds A[l][ka] in register ka
                // ---- exchange: lane ka receives A[n1][ka] for all n1 -- real parts, then imaginary parts, through the wave's rows
                {
                    float re[64];
#pragma unroll
                    for (int ka = 0; ka < 64; ++ka) rows[ka * kW64Row + lane] = v[ka].x;
                    wave_sync();
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const float4 r4 = reinterpret_cast<const float4 *>(rows + lane * kW64Row)[q];
                        re[4 * q] = r4.x;  re[4 * q + 1] = r4.y;  re[4 * q + 2] = r4.z;  re[4 * q + 3] = r4.w;
                    }
                    wave_sync();
#pragma unroll
                    for (int ka = 0; ka < 64; ++ka) rows[ka * kW64Row + lane] = v[ka].y;
                    wave_sync();
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const float4 r4 = reinterpret_cast<const float4 *>(rows + lane * kW64Row)[q];
                        v[4 * q] = cf{re[4 * q], r4.x};          v[4 * q + 1] = cf{re[4 * q + 1], r4.y};
                        v[4 * q + 2] = cf{re[4 * q + 2], r4.z};  v[4 * q + 3] = cf{re[4 * q + 3], r4.w};
                    }
                    wave_sync();
                }
                // ---- twiddles W_N^(ka n1), n1 = c + 8 d: W^(8 ka d) (rows 0..6) times W^(ka c) (rows 7..13)
#pragma unroll
                for (int c = 1; c < 8; ++c) {
                    const cf wc = tw[(6 + c) * 64 + lane];
                    v[c] = cmul(v[c], wc);
                }
#pragma unroll
                for (int d = 1; d < 8; ++d) {
                    const cf wd = tw[(d - 1) * 64 + lane];
                    v[8 * d] = cmul(v[8 * d], wd);
#pragma unroll
                    for (int c = 1; c < 8; ++c) {
                        const cf wc = tw[(6 + c) * 64 + lane];
                        v[c + 8 * d] = cmul(cmul(v[c + 8 * d], wd), wc);
                    }
                }
                // ---- pass 2 (over n1): X[ka + 64 kb] in register kb
                dft64(v);
                if constexpr (LIN) {
                    // X[k] -= (sum x) W[k] / N for k in {0, 1, N - 1}: the constant detrend, applied to the transform
                    // (FFT(w (x - m)) = FFT(w x) - m W, W real and confined to those bins for a cosine-sum window of order <= 1)
                    const float k0 = lane == 0 ? p.lin_c[0] : lane == 1 ? p.lin_c[1] : 0.f;
                    const float k63 = lane == 63 ? p.lin_c[2] : 0.f;
                    v[0].x = __builtin_fmaf(-k0, sum.x, v[0].x);
                    v[0].y = __builtin_fmaf(-k0, sum.y, v[0].y);
                    v[63].x = __builtin_fmaf(-k63, sum.x, v[63].x);
                    v[63].y = __builtin_fmaf(-k63, sum.y, v[63].y);
                }
                // |X|^2 * scale (scipy _spectral_py.py:2126-2128); sqrt(scale) is folded into the window table
                float P[64];
#pragma unroll
                for (int r = 0; r < 64; ++r) P[r] = __builtin_fmaf(v[r].x, v[r].x, v[r].y * v[r].y);

                if constexpr (SUMS) {
                    if (!halo) {
#pragma unroll
                        for (int r = 0; r < 64; ++r) acc[r] += P[r];
                    }
                }
                // spectrogram row (dense modes) and look-back tail column (the last K segments): bin = lane + 64 r
                if constexpr (MODE == 1 || MODE == 2) {
                    const rsrc_t rs = make_rsrc(p.spec + ((int64_t)s_pos * T + seg) * N, (uint32_t)(N * sizeof(float)));
#pragma unroll
                    for (int r = 0; r < 64; ++r) raw_buffer_store_f1(P[r], rs, lane * 4, 256 * r, 0);
                }
                if constexpr (SUMS && !(RT_W64_ABL & 2)) {
                    const int col = seg - (T - p.tail_cols);
                    if (!halo && col >= 0) {
                        // Sparse tail (the scans that keep threshold bits): the next buffer's look-back walks down from the last
                        // segment while the cells pass the absolute threshold and stops ON the first that does not, so it can
                        // only reach a cell whose later cells are all hot.  A cell is written iff the later cells OF ITS CHUNK
                        // are (`allhot` before this step's update, all ones at the chunk's last segment): a superset.
                        const rsrc_t rt_ = make_rsrc(p.tail + ((int64_t)s * p.tail_cols + col) * N, (uint32_t)(N * sizeof(float)));
                        if (!FLAGS || __builtin_amdgcn_ballot_w64((allhot.lo & allhot.hi) != ~0u) == 0ull) {
#pragma unroll
                            for (int r = 0; r < 64; ++r) raw_buffer_store_f1(P[r], rt_, lane * 4, 256 * r, 0);
                        } else {
                            const uint32_t any_lo = wave_or(allhot.lo), any_hi = wave_or(allhot.hi);
#pragma unroll
                            for (int r = 0; r < 64; ++r) {
                                if (bit_of(any_lo, any_hi, r)) {  // (scalar branch: some lane writes this register's cell)
                                    if (bit_of(allhot, r)) raw_buffer_store_f1(P[r], rt_, lane * 4, 256 * r, 0);
                                }
                            }
                        }
                    }
                }

                if constexpr (TEST) {
                    // candidates are rare: one max over the lane's 64 cells and a single compare in the common path (a NaN
                    // cell means the whole segment is NaN, so the max is, and `!(m < thr)` holds as for the reference's
                    // `not (P < thr)`)
                    float mx = __builtin_fmaxf(__builtin_fmaxf(P[0], P[1]), P[2]);
#pragma unroll
                    for (int r = 3; r < 63; r += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, P[r]), P[r + 1]);
                    mx = __builtin_fmaxf(mx, P[63]);
                    bits64 hot{0u, 0u};
                    if (!LISTED && !(mx < thr)) {
#pragma unroll
                        for (int r = 31; r >= 0; --r) {
                            hot.lo = (hot.lo << 1) | ((P[r] < thr) ? 0u : 1u);
                            hot.hi = (hot.hi << 1) | ((P[r + 32] < thr) ? 0u : 1u);
                        }
                    }
                    if constexpr (FLAGS) {
                        if (!halo) {
                            if constexpr (MODE == 6) {
                                bits64 bits = hot;
                                if (p.thr_bin && any(bits)) {
                                    // the bin's own second threshold (a lower bound of snr * row mean, make_bin_thresholds)
                                    const float4 *tb = reinterpret_cast<const float4 *>(p.thr_bin + ((int64_t)s * 64 + lane) * 64);
                                    bits64 ok{0u, 0u};
#pragma unroll
                                    for (int q = 7; q >= 0; --q) {
                                        const float4 a4 = tb[q], b4 = tb[q + 8];
                                        ok.lo = (ok.lo << 1) | ((P[4 * q + 3] < a4.w) ? 0u : 1u);
                                        ok.lo = (ok.lo << 1) | ((P[4 * q + 2] < a4.z) ? 0u : 1u);
                                        ok.lo = (ok.lo << 1) | ((P[4 * q + 1] < a4.y) ? 0u : 1u);
                                        ok.lo = (ok.lo << 1) | ((P[4 * q + 0] < a4.x) ? 0u : 1u);
                                        ok.hi = (ok.hi << 1) | ((P[32 + 4 * q + 3] < b4.w) ? 0u : 1u);
                                        ok.hi = (ok.hi << 1) | ((P[32 + 4 * q + 2] < b4.z) ? 0u : 1u);
                                        ok.hi = (ok.hi << 1) | ((P[32 + 4 * q + 1] < b4.y) ? 0u : 1u);
                                        ok.hi = (ok.hi << 1) | ((P[32 + 4 * q + 0] < b4.x) ? 0u : 1u);
                                    }
                                    bits = bits & ok;
                                }
                                store_bits(p.cell_hot, ((int64_t)s * T + seg) * 64 + lane, bits);
                            }
                            if constexpr (MODE == 4 || MODE == 6) n_abs += (uint32_t)(__builtin_popcount(hot.lo) + __builtin_popcount(hot.hi));
                            allhot = allhot & hot;
                            if (chunk == 0 && p.full) store_bits(p.first, ((int64_t)s * L + seg) * 64 + lane, hot);
                        }
                    }
                    if constexpr (EMIT) {
                        // a cell is kept if it is a candidate itself or directly precedes one (T11)
                        const bits64 need_seg = need | first_nxt;
                        const bits64 emit = LISTED ? need_seg : (halo ? (next_hot & ~hot) : (hot | next_hot)) & need_seg;
                        if (!(RT_W64_ABL & 32) && !gave_up && __builtin_amdgcn_ballot_w64(any(emit)) != 0ull) {  // wave-uniform, rare
                            // Candidates are staged per wave in LDS and flushed with ONE returned atomic per bucket and
                            // flush.  Each lane stages its own cells: the powers go to the (free) exchange rows first, so
                            // that a lane can pick the registers its bits name in a short loop -- visiting the 64 registers
                            // with scalar branches on the wave-wide union of the words cost 0.6 ms of a 5.8-ms launch
                            // (profiles/r04_c_*: taken branches and SALU chains, not the atomics).
#pragma unroll
                            for (int r = 0; r < 64; ++r) rows[r * kW64Row + lane] = P[r];
                            wave_sync();
                            // exclusive prefix of the lanes' cell counts, bit plane by bit plane (mbcnt of a ballot), and their total
                            const uint32_t cnt = (uint32_t)(__builtin_popcount(emit.lo) + __builtin_popcount(emit.hi));
                            uint32_t before = 0;
                            int cells = 0;
#pragma unroll
                            for (int b = 0; b < 7; ++b) {
                                const unsigned long long m = __builtin_amdgcn_ballot_w64((cnt >> b) & 1u);
                                before += (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0)) << b;
                                cells += __builtin_popcountll(m) << b;
                            }
                            if (stg_n + cells > kW64Stage) {
                                flush_stage64(p, s, rows, stg_n);
                                stg_n = 0;
                            }
                            const uint32_t key0 = ((uint32_t)lane << p.tbits) | (uint32_t)seg;
                            const bool staged = cells <= kW64Stage;
                            if (!staged) {
                                // more than a staging area in one step (dense input): straight to memory -- unless one of
                                // the stream's lists has overflowed already (count > capacity): then the call is re-run on
                                // another level (AUTO) or fails (SPARSE) whatever else is emitted
                                const uint32_t have = lane < kBuckets ? __hip_atomic_load(&p.hot_count[s * kBuckets + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                                gave_up = __builtin_amdgcn_ballot_w64(have > (uint32_t)p.hot_cap) != 0ull;
                            }
                            if (!gave_up) {
                                const int bkt = lane & (kBuckets - 1);
                                uint32_t lo = emit.lo, hi = emit.hi;
                                int idx = stg_n + (int)before;
#pragma nounroll
                                while ((lo | hi) != 0u) {  // (per lane: as many rounds as the busiest lane has cells)
                                    int r;
                                    if (lo) { r = __builtin_ctz(lo); lo &= lo - 1u; } else { r = 32 + __builtin_ctz(hi); hi &= hi - 1u; }
                                    const uint2 cell = make_uint2(key0 + ((uint32_t)(64 * r) << p.tbits), __float_as_uint(rows[r * kW64Row + lane]));
                                    if (staged) {
                                        *w64_stage_cell(rows, idx++) = cell;
                                    } else {
                                        const uint32_t slot = atomicAdd(&p.hot_count[s * kBuckets + bkt], 1u);
                                        if (slot < (uint32_t)p.hot_cap) p.hot[((int64_t)s * kBuckets + bkt) * p.hot_cap + slot] = cell;
                                    }
                                }
                                if (staged) stg_n += cells;
                            }
                            wave_sync();  // (the rows are the next step's exchange area)
                        }
                    }
                    next_hot = hot;
                }

                // ---- the next step's segment
                if (halo) break;
                if constexpr (LISTED) {
                    if (seg_after < 0) break;
                    seg = seg_after;
                    ++k7;
                } else if (seg > c0) {
                    --seg;
                } else {
                    if constexpr (EMIT) {
                        if (!(RT_W64_ABL & 16) && c0 > 0 && __builtin_amdgcn_ballot_w64(any(next_hot & need)) != 0ull) {  // a lowest cell of the chunk is a candidate
                            seg = c0 - 1;
                            halo = true;
                            continue;
                        }
                    }
                    break;
                }
            }

            if constexpr (EMIT) {
                if (stg_n) flush_stage64(p, s, rows, stg_n);
            }
            if constexpr (FLAGS) {
                if (p.full) store_bits(p.full, ((int64_t)s * p.chunks + chunk) * 64 + lane, allhot);
            }
            if constexpr (MODE == 4 || MODE == 6) {
                if (p.abs_hot) {
                    uint32_t n = n_abs;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) n += (uint32_t)__shfl_xor((int)n, o);
                    if (lane == 0 && n) atomicAdd(&p.abs_hot[s], n);
                }
            }
            if constexpr (MODE == 3) {
                if (acc[0] == 12345.678f) p.psum[0] = acc[0];  // keeps the loads alive, never true in practice
            } else if constexpr (SUMS) {
                // one partial row of sums per item (= chunk): the detection adds a stream's rows in chunk order, float64
                const rsrc_t rp = make_rsrc(p.psum + ((int64_t)s * p.blocks_per_stream + chunk) * N, (uint32_t)(N * sizeof(float)));
#pragma unroll
                for (int r = 0; r < 64; ++r) raw_buffer_store_f1(acc[r], rp, lane * 4, 256 * r, 0);
                if (p.chunk_min && (c0 + L <= T)) {
                    // the quietest complete chunk of the bin so far (positive floats order like their bits): make_bin_thresholds
                    uint32_t *cm = p.chunk_min + (int64_t)s * N + lane;
#pragma unroll
                    for (int r = 0; r < 64; ++r) atomicMin(cm + 64 * r, __float_as_uint(acc[r]));
                }
            }
        } while (false);
        item = n_waves + __builtin_amdgcn_readfirstlane((int)ticket);
    }
    // the wave that leaves last puts the counters back for the next launch on this stream
    if (lane == 0) {
        const uint32_t left = atomicAdd(p.work + 1, 1u);
        if (left + 1u == (uint32_t)n_waves) {
            atomicExch(p.work, 0u);
            atomicExch(p.work + 1, 0u);
        }
    }
}

}  // namespace rt
#endif
