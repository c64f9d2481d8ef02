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
//     n = l + 64 m        lane l holds x[l + 64 m], m = 0 .. 63        (a 64-sample piece m is 512 consecutive bytes)
//     pass 1 (in-lane)    A[l][ka] = sum_m w x[l + 64 m] W64^(m ka)
//     exchange            lane ka receives A[n1][ka], n1 = 0 .. 63      (wave-private 64 x 64 transpose in LDS, real parts then
//                                                                        imaginary parts through the same 16 KiB; no barrier)
//     twiddle             x W_N^(ka n1),  n1 = c + 8 d:  W^(8 ka d) x W^(ka c), fourteen table rows in LDS
//     pass 2 (in-lane)    X[ka + 64 kb] = sum_n1 ... W64^(n1 kb)  ->  register kb of lane ka holds bin ka + 64 kb
// so bins are in natural order across lanes (every store of a spectrogram row or tail column is 256 consecutive bytes, no
// LDS staging), a lane's candidate bucket is lane & 15, and nothing a wave does depends on another wave: the only workgroup
// barrier of the kernel is the one behind the table staging.
//
// Memory.  64 points per lane + 64 row sums leave no registers for a second segment, so half of the NEXT segment travels
// through LDS instead: as soon as a step's exchange is over, its 16 KiB take the next segment's quarters 0 and 1 (the pieces
// m = 4 j, 4 j + 1: 1 KiB runs, sixteen `buffer_load_dwordx4 ... lds` -- no registers, no instructions to move the data on)
// while the wave goes on with the twiddles, pass 2 and the threshold test.  The next step asks for quarters 2 and 3 by
// ordinary loads (32 instead of 64: a vector-memory instruction costs the issuing wave ~50 cycles), reads quarters 0 and 1
// from LDS and transforms them while 2 and 3 arrive.  The exchange rows carry no pad for that (a DMA piece is written as it
// lies): element (n1, ka) sits in row ka at the 16-byte chunk (n1 / 4) ^ (ka & 15), conflict-free for the column stores and
// the row loads alike.  Two waves per SIMD cover what is left of each other's HBM round trips.
//
// One workgroup of eight waves per CU: 8 x (16 KiB exchange / prefetch area + 1 KiB candidate staging: 128 cells) + the window
// table (16 KiB, in the order the lanes read it) + 16 twiddle rows = 163 840 B, all of a CU's LDS.
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
constexpr int kW64AreaFloats = 64 * 64;               // one wave's exchange rows (XOR-swizzled, no pad) = half a segment of prefetched samples
constexpr int kW64Stage = 128;                        // candidate cells staged per wave before a flush (1 KiB behind the area)
constexpr int kW64XchFloats = kW64AreaFloats + 2 * kW64Stage;  // one wave's LDS
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

// LDS-DMA: a buffer load whose data goes straight to LDS at (wave-uniform base) + lane * size, no destination registers
__device__ void raw_buffer_load_lds(rsrc_t rsrc, __attribute__((address_space(3))) void *lds, int size, int voffset, int soffset, int offset, int aux)
    __asm("llvm.amdgcn.raw.buffer.load.lds");

// Quarters 0 and 1 of a segment (pieces m = 4 j and 4 j + 1: runs of two pieces, 1 KiB of complex64 / 256 B of uint8 pairs)
// into a wave's area, piece (2 j + n0) at element offset 64 (2 j + n0): sixteen instructions, nothing to wait for here.
template <class raw_t>
__device__ __forceinline__ void w64_prefetch(rsrc_t r, float *area, int lane) {
    constexpr int PB = 64 * (int)sizeof(raw_t);  // bytes of a piece
    constexpr int SZ = 2 * PB / 64;              // bytes per lane and instruction
#pragma unroll
    for (int j = 0; j < 16; ++j)
        raw_buffer_load_lds(r, (__attribute__((address_space(3))) void *)(reinterpret_cast<char *>(area) + 2 * PB * j), SZ, lane * SZ, 4 * PB * j, 0, kAuxNT);
}

// P[r] for a register index that differs from lane to lane (rt_kernels.h: pick_range, 63 v_cndmask)
__device__ __forceinline__ float pick64(const float (&P)[64], int r) { return pick_range<0, 64>(P, r); }

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

// the transform's arithmetic form (rt_fft.h): 1 = packed pairs (v_pk_add / v_pk_fma_f32), 0 = scalar -- the same results
#ifndef RT_W64_PK
#define RT_W64_PK 0
#endif
// 0 = quarters 0 and 1 are asked for at the start of their own step (diagnostic: the same instructions without the lead)
#ifndef RT_W64_PREFETCH
#define RT_W64_PREFETCH 1
#endif
// 1 (round 6) = quarter 2 of the NEXT segment is requested into registers as soon as a step's powers are taken -- the transform's 128
// registers are dead from there on --, so that the next step asks for quarter 3 alone and its pass 1 waits for one late quarter instead
// of two (stage stamps, round 4: 5.9 k + 1.8 k of a step's 21.5 k cycles went into pass 1 waiting for the second half).  0 = both
// quarters at the head of their own step (diagnostic builds: the A/B).
#ifndef RT_W64_Q2AHEAD
#define RT_W64_Q2AHEAD 0  // (the product: see EXPERIMENTS.md, round 6 -- with the sixteen pairs carried over the loop edge hipcc spills 45 - 55 registers inside the step)
#endif
// (Tried and dropped: spreading a step's vector-memory instructions over its arithmetic -- quarter 3's loads behind quarter 0's
// transform, the prefetch pieces two at a time between the twiddle rows -- because a wave that issues them in one run waits for
// queue slots in between (38 cycles per load, 76 per LDS-DMA piece, profiles/r04_c_stage_stamps_*).  Either placement sends hipcc's
// register allocation from 2 spilled registers outside the step to 53 / 152 inside it.)


template <int MODE, bool U8, bool LIN>
__global__ __launch_bounds__(kW64Block, 1) void stft_scan64(const StftParams p) {
    using raw_t = typename std::conditional<U8, iq_u8, cf>::type;
    using C = typename std::conditional<RT_W64_PK == 1, cfv, cf>::type;    // pass 1
    using C2 = typename std::conditional<RT_W64_PK != 0, cfv, cf>::type;   // twiddles and pass 2
    constexpr int N = 4096;
    constexpr bool EMIT = (MODE == 0 || MODE == 5 || MODE == 7);   // candidate cells go to the bucket lists
    constexpr bool FLAGS = (MODE == 0 || MODE == 4 || MODE == 6);  // threshold bits are kept
    constexpr bool SUMS = (MODE != 2 && MODE != 3 && MODE != 5 && MODE != 7);  // row sums and look-back tail
    constexpr bool LISTED = (MODE == 7);                           // the steps take the segments plan_runs listed
    constexpr bool TEST = (EMIT || FLAGS) && !(RT_W64_ABL & 1);       // the step ends with the threshold test

    __shared__ __attribute__((aligned(16))) float lds[kW64LdsBytes / sizeof(float)];
    // (the tables first: a DS instruction's immediate offset reaches 64 KiB, so every row of a table below that mark is the
    // lane's one address register + an immediate -- behind the waves' areas each row cost a register of its own)
    cf *const tw = reinterpret_cast<cf *>(lds);                                         // [16][64]
    float4 *const win = reinterpret_cast<float4 *>(lds + 2 * kW64TwRows * 64);          // [n0][j / 4][lane]
    float *const areas = lds + 2 * kW64TwRows * 64 + 4096;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (scalar: the wave's LDS addresses stay in SGPRs -- an LDS-DMA takes its destination from M0)
    for (int i = tid; i < kW64TwRows * 64; i += kW64Block) tw[i] = p.tw1[i];
    for (int i = tid; i < 1024; i += kW64Block) win[i] = reinterpret_cast<const float4 *>(p.window_t)[i];
    __syncthreads();  // the kernel's only workgroup barrier
    float *const area = areas + wave * kW64XchFloats;                               // exchange rows / prefetched half segment
    uint2 *const stg = reinterpret_cast<uint2 *>(area + kW64AreaFloats);            // this wave's candidate staging
    // which segment's quarters 0 and 1 are in the area or on their way there (stream, segment; -1: none)
    int pre_s = -1, pre_seg = -1;

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
            if (LIN && p.sub_first && p.sub_first[s] != 0) break;  // a stream the guard of this form has marked: the subtract-first launch behind this one analyses it
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
            const ChunkSpan span = chunk_span(chunk, L, p.short_chunks, p.short_len);  // (the earliest chunks -- the last items -- may be short)
            const int c0 = span.c0, c_len = span.len;
            const float thr = p.thr_s ? p.thr_s[s] : p.thr;  // wave-uniform
            const raw_t *stream_iq = reinterpret_cast<const raw_t *>(p.iq) + (int64_t)s * p.stream_stride;

#ifdef RT_STAMPS  // diagnostic build: cycles per stage of the step, summed over the item's steps (rt_kernels.h: RT_STAMP)
            uint32_t st_acc[kStamps] = {};
            const uint32_t st_wave_start = (uint32_t)__builtin_amdgcn_s_memrealtime();
            uint32_t st_prev = (uint32_t)__builtin_amdgcn_s_memtime();
            const uint32_t st_t0 = st_prev, st_r0 = (uint32_t)__builtin_amdgcn_s_memrealtime();
            uint32_t st_steps = 0;
#endif
            float acc[64];
#pragma unroll
            for (int r = 0; r < 64; ++r) acc[r] = 0.f;
            bits64 next_hot{0u, 0u};            // threshold bits of the segment one later in time
            bits64 allhot{~0u, ~0u};            // FLAGS: the chunk's bits so far
            bits64 need{~0u, ~0u};              // MODE 5: the lane's bins that may emit in every segment of the chunk
            uint32_t n_abs = 0;                 // MODE 4 / 6: this lane's cells at or above the absolute threshold
            float dc_acc = 0.f;                 // LIN: sum over the item's segments of |sum of the segment's samples|^2 (wave-uniform)
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
            int seg = LISTED ? p.seg_list[(int64_t)s * T + e0] : ((c0 + c_len < T ? c0 + c_len : T) - 1);
            bool halo = false;
            // RT_W64_Q2AHEAD: quarter 2 of a step's segment is requested by the step BEFORE (every step knows its successor: the chunk
            // walks down, a listed pass reads its list one entry ahead, the segment below the chunk is asked for in any case) -- and
            // for the item's first step here, so that the step loop has ONE place that defines these sixteen registers.
            raw_t q2r[16];
            if constexpr (RT_W64_Q2AHEAD != 0 && MODE != 3) {
                const rsrc_t r0 = make_rsrc(stream_iq + (int64_t)seg * N, (uint32_t)(N * sizeof(raw_t)));
#pragma unroll
                for (int j = 0; j < 16; ++j) q2r[j] = buf_load_iq(r0, lane * (int)sizeof(raw_t), 64 * (2 + 4 * j) * (int)sizeof(raw_t), raw_t{});
            }
            for (;;) {
                RT_STAMP(0);  // loop control, the previous step's threshold test and emission
                bits64 first_nxt{0u, 0u};  // MODE 5, chunk 0: the bins whose run through t = 0 reaches this segment; MODE 7: the cells to emit
                if constexpr (MODE == 5) {
                    if (chunk == 0 && seg < L) first_nxt = load_bits(p.first, ((int64_t)s * L + seg) * 64 + lane);
                }
                int seg_after = -1;
                if constexpr (LISTED) {
                    first_nxt = load_bits(p.cell_need, ((int64_t)s * T + seg) * 64 + lane);
                    if (k7 + 1 < n_mine) seg_after = p.seg_list[(int64_t)s * T + e0 + k7 + 1];
                }

                // ---- the segment's samples.  Quarter n0 = the elements m = n0 + 4 j; quarters 0 and 1 come through LDS (asked for
                // during the previous step where that was possible), 2 and 3 by loads issued now.
                C v[64];
                cf sum{0.f, 0.f};  // of the raw samples, the same bits in every lane (detrend='constant', scipy _signaltools.py:3926)
                {
                    const rsrc_t r = make_rsrc(stream_iq + (int64_t)seg * N, (uint32_t)(N * sizeof(raw_t)));
                    if (!RT_W64_PREFETCH || pre_s != s || pre_seg != seg) {
                        // (first step of an item, or a step the previous one could not foresee)
                        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): a stale prefetch has landed before this one is written over it
                        w64_prefetch<raw_t>(r, area, lane);
                    }
                    pre_seg = -1;  // (this step's exchange writes over the area: what it held is gone unless the step asks for more)
                    // quarters 2 and 3, sixteen registers each: element [j].  (Quarter 2 lives in the SAME sixteen registers whether the
                    // step before requested it -- older than the loads below, in flight or here -- or this one does.)
                    if constexpr (!(RT_W64_Q2AHEAD != 0 && MODE != 3)) {
#pragma unroll
                        for (int j = 0; j < 16; ++j) q2r[j] = buf_load_iq(r, lane * (int)sizeof(raw_t), 64 * (2 + 4 * j) * (int)sizeof(raw_t), raw_t{});
                    }
                    raw_t q3r[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) q3r[j] = buf_load_iq(r, lane * (int)sizeof(raw_t), 64 * (3 + 4 * j) * (int)sizeof(raw_t), raw_t{});
                    RT_STAMP(1);  // issue: a prefetch nobody made + the 32 loads of quarters 2 and 3
                    __builtin_amdgcn_sched_barrier(0);
                    // the sixteen LDS-DMA pieces are older than the 32 loads: all but the 32 youngest operations done = the area is filled
                    __builtin_amdgcn_s_waitcnt(0x8F70);  // vmcnt(32)
                    __builtin_amdgcn_sched_barrier(0);
                    RT_STAMP(2);  // wait for the prefetched half
                    // (lane indices made opaque inside the step: otherwise hipcc hoists the lanes' LDS addresses -- one per exchange
                    // chunk, table row and prefetched piece, 54 of them -- out of the step loop, where they cost the registers the
                    // row sums need: spills in every step)
                    int lane_a = lane;
                    asm volatile("" : "+v"(lane_a));
                    const raw_t *pre = reinterpret_cast<const raw_t *>(area);
                    auto sample = [&](int n0, int j) -> cf {  // x[lane + 64 (n0 + 4 j)]
                        return to_cf(n0 < 2 ? pre[(2 * j + n0) * 64 + lane_a] : n0 == 2 ? q2r[j] : q3r[j]);
                    };
                    if constexpr (MODE == 3) {
                        // traffic calibration: the scan's exact load stream, nothing else
#pragma unroll
                        for (int n0 = 0; n0 < 4; ++n0)
#pragma unroll
                            for (int j = 0; j < 16; ++j) {
                                const cf x = sample(n0, j);
                                acc[0] += x.x + x.y;
                            }
                        wave_sync();
                        if (seg > c0) {
                            --seg;
                            if (RT_W64_PREFETCH) {
                                w64_prefetch<raw_t>(make_rsrc(stream_iq + (int64_t)seg * N, (uint32_t)(N * sizeof(raw_t))), area, lane);
                                pre_s = s;
                                pre_seg = seg;
                            }
                            continue;
                        }
                        break;
                    }
                    // pass 1 (over m), a quarter at a time: window, 16-point transform over j.  The sum of the raw samples is taken
                    // as they come (four running sums per quarter, one fixed order), so that a sample's registers are free once
                    // its windowed value exists.
                    cf q[4];
                    if constexpr (LIN) {
                        // constant detrend by linearity: the sum is only needed after pass 2 (see below)
#pragma unroll
                        for (int n0 = 0; n0 < 4; ++n0) {
                            C a[16];
                            cf part[4] = {cf{0.f, 0.f}, cf{0.f, 0.f}, cf{0.f, 0.f}, cf{0.f, 0.f}};
#pragma unroll
                            for (int jq = 0; jq < 4; ++jq) {
                                const float4 w4 = win[(n0 * 4 + jq) * 64 + lane_a];
                                const float w[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const cf x = sample(n0, 4 * jq + e);
                                    if (!(RT_W64_ABL & 4)) part[e] = cadd(part[e], x);
                                    const cf y = cscale(x, w[e]);
                                    a[4 * jq + e] = make_c<C>(y.x, y.y);
                                }
                            }
                            q[n0] = cadd(cadd(part[0], part[1]), cadd(part[2], part[3]));
                            dft16(a);
#pragma unroll
                            for (int j = 0; j < 16; ++j) v[n0 + 4 * j] = a[j];  // A[n0][k' = j]
                            __builtin_amdgcn_sched_barrier(0);  // (quarter by quarter: the reads and transforms of later quarters stay behind)
                        }
                        sum = wave_sum<64>(cadd(cadd(q[0], q[2]), cadd(q[1], q[3])));
                    } else {
                        // subtract-first form: the mean is needed before the window, i.e. once the whole segment has arrived
                        cf x[4][16];
#pragma unroll
                        for (int n0 = 0; n0 < 4; ++n0) {
                            cf part[4] = {cf{0.f, 0.f}, cf{0.f, 0.f}, cf{0.f, 0.f}, cf{0.f, 0.f}};
#pragma unroll
                            for (int j = 0; j < 16; ++j) {
                                x[n0][j] = sample(n0, j);
                                part[j & 3] = cadd(part[j & 3], x[n0][j]);
                            }
                            q[n0] = cadd(cadd(part[0], part[1]), cadd(part[2], part[3]));
                        }
                        sum = wave_sum<64>(cadd(cadd(q[0], q[2]), cadd(q[1], q[3])));
                        const cf mean = cscale(sum, 1.0f / (float)N);
#pragma unroll
                        for (int n0 = 0; n0 < 4; ++n0) {
                            C a[16];
#pragma unroll
                            for (int jq = 0; jq < 4; ++jq) {
                                const float4 w4 = win[(n0 * 4 + jq) * 64 + lane_a];
                                const float w[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const cf y = cscale(csub(x[n0][4 * jq + e], mean), w[e]);
                                    a[4 * jq + e] = make_c<C>(y.x, y.y);
                                }
                            }
                            dft16(a);
#pragma unroll
                            for (int j = 0; j < 16; ++j) v[n0 + 4 * j] = a[j];
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
                RT_STAMP(3);  // pass 1: LDS reads, waits for quarters 2 and 3, window, four 16-point transforms
                dft64_finish(v);  // lane l now holds A[l][ka] in register ka
                RT_STAMP(4);  // pass 1: twiddles and 4-point transforms
                // ---- exchange: lane ka receives A[n1][ka] for all n1 -- real parts, then imaginary parts, through the wave's area.
                // Element (n1, ka) lies in row ka (256 B) at the 16-byte chunk (n1 / 4) ^ (ka & 15): lane l = n1 stores at byte
                // (4 l) ^ (16 (ka & 15)) of the row, lane ka loads chunk q of its row from byte (256 ka + 16 (ka & 15)) ^ (16 q).
                C2 u[64];
                {
                    char *const ab = reinterpret_cast<char *>(area);
                    int lane_x = lane;
                    asm volatile("" : "+v"(lane_x));
                    const int st0 = lane_x * 4, ld0 = lane_x * 256 + ((lane_x & 15) << 4);
                    float re[64];
                    wave_sync();  // (the prefetched samples have been read)
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int off = st0 ^ (k << 4);
#pragma unroll
                        for (int g = 0; g < 4; ++g) *reinterpret_cast<float *>(ab + (k + 16 * g) * 256 + off) = v[k + 16 * g].x;
                    }
                    wave_sync();
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const float4 r4 = *reinterpret_cast<const float4 *>(ab + (ld0 ^ (q << 4)));
                        re[4 * q] = r4.x;  re[4 * q + 1] = r4.y;  re[4 * q + 2] = r4.z;  re[4 * q + 3] = r4.w;
                    }
                    wave_sync();
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int off = st0 ^ (k << 4);
#pragma unroll
                        for (int g = 0; g < 4; ++g) *reinterpret_cast<float *>(ab + (k + 16 * g) * 256 + off) = v[k + 16 * g].y;
                    }
                    wave_sync();
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const float4 r4 = *reinterpret_cast<const float4 *>(ab + (ld0 ^ (q << 4)));
                        u[4 * q] = make_c<C2>(re[4 * q], r4.x);          u[4 * q + 1] = make_c<C2>(re[4 * q + 1], r4.y);
                        u[4 * q + 2] = make_c<C2>(re[4 * q + 2], r4.z);  u[4 * q + 3] = make_c<C2>(re[4 * q + 3], r4.w);
                    }
                    wave_sync();
                }
                RT_STAMP(5);  // exchange
                // ---- the area is free: quarters 0 and 1 of the segment the next step will take, where this step can tell.  At a
                // chunk's lowest segment that is the segment below it, needed only if a lowest cell turns out hot (half the chunks
                // of config 5): asked for anyway, 16 KiB per chunk against the chunk's L x 32 KiB.
                int nxt = -1;  // the segment the next step will (or may) take, where this step can tell
                if (RT_W64_PREFETCH) {
                    if constexpr (LISTED) nxt = seg_after;
                    else if (!halo) nxt = (seg > c0) ? seg - 1 : ((EMIT && c0 > 0) ? c0 - 1 : -1);
                    if (nxt >= 0) {
                        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the exchange's loads have returned
                        w64_prefetch<raw_t>(make_rsrc(stream_iq + (int64_t)nxt * N, (uint32_t)(N * sizeof(raw_t))), area, lane);
                        pre_s = s;
                        pre_seg = nxt;
                    }
                }
                RT_STAMP(6);  // issue of the next segment's prefetch
                // ---- twiddles W_N^(ka n1), n1 = c + 8 d: W^(8 ka d) (rows 0..6) times W^(ka c) (rows 7..13)
                int lane_t = lane;
                asm volatile("" : "+v"(lane_t));
#pragma unroll
                for (int c = 1; c < 8; ++c) {
                    const cf wc_ = tw[(6 + c) * 64 + lane_t];
                    const C2 wc = make_c<C2>(wc_.x, wc_.y);
                    u[c] = cmul(u[c], wc);
                }
#pragma unroll
                for (int d = 1; d < 8; ++d) {
                    const cf wd_ = tw[(d - 1) * 64 + lane_t];
                    const C2 wd = make_c<C2>(wd_.x, wd_.y);
                    u[8 * d] = cmul(u[8 * d], wd);
#pragma unroll
                    for (int c = 1; c < 8; ++c) {
                        const cf wc_ = tw[(6 + c) * 64 + lane_t];
                        const C2 wc = make_c<C2>(wc_.x, wc_.y);
                        u[c + 8 * d] = cmul(cmul(u[c + 8 * d], wd), wc);
                    }
                }
                RT_STAMP(7);  // twiddles
                // ---- pass 2 (over n1): X[ka + 64 kb] in register kb
                dft64(u);
                RT_STAMP(8);  // pass 2
                if constexpr (LIN) {
                    if constexpr (SUMS) {
                        if (!halo) dc_acc = __builtin_fmaf(sum.x, sum.x, __builtin_fmaf(sum.y, sum.y, dc_acc));  // (guard of this form: StftParams::dc_flag)
                    }
                    // X[k] -= (sum x) W[k] / N for k in {0, 1, N - 1}: the constant detrend, applied to the transform
                    // (FFT(w (x - m)) = FFT(w x) - m W, W real and confined to those bins for a cosine-sum window of order <= 1)
                    const float k0 = lane == 0 ? p.lin_c[0] : lane == 1 ? p.lin_c[1] : 0.f;
                    const float k63 = lane == 63 ? p.lin_c[2] : 0.f;
                    u[0].x = __builtin_fmaf(-k0, sum.x, u[0].x);
                    u[0].y = __builtin_fmaf(-k0, sum.y, u[0].y);
                    u[63].x = __builtin_fmaf(-k63, sum.x, u[63].x);
                    u[63].y = __builtin_fmaf(-k63, sum.y, u[63].y);
                }
                // |X|^2 * scale (scipy _spectral_py.py:2126-2128); sqrt(scale) is folded into the window table
                float P[64];
#pragma unroll
                for (int r = 0; r < 64; ++r) P[r] = __builtin_fmaf(u[r].x, u[r].x, u[r].y * u[r].y);
                if constexpr (SUMS) {
                    if (!halo) {
#pragma unroll
                        for (int r = 0; r < 64; ++r) acc[r] += P[r];
                    }
                }
                // spectrogram row (dense modes) and look-back tail column (the last K segments): bin = lane + 64 r
                if constexpr (MODE == 1 || MODE == 2) {
                    const rsrc_t rs = make_rsrc(p.spec + ((int64_t)(p.spec_by_stream ? s : s_pos) * T + seg) * N, (uint32_t)(N * sizeof(float)));
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

                RT_STAMP(9);  // detrend correction, power, row sums, spectrogram row / tail column
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
                        if constexpr (LISTED) {
                            // consumed: back to zero (the planner writes only the words that keep anything, rt_kernels.h: plan_runs)
                            if (first_nxt.lo | first_nxt.hi) store_bits(const_cast<uint16_t *>(p.cell_need), ((int64_t)s * T + seg) * 64 + lane, bits64{0u, 0u});
                        }
                        const bits64 emit = LISTED ? need_seg : (halo ? (next_hot & ~hot) : (hot | next_hot)) & need_seg;
                        if (!(RT_W64_ABL & 32) && !gave_up && __builtin_amdgcn_ballot_w64(any(emit)) != 0ull) {  // wave-uniform, rare
                            // Candidates are staged per wave in LDS and flushed with ONE returned atomic per bucket and
                            // flush.  Each lane stages its own cells, in a short loop over the bits of its word; the power of
                            // the register a bit names comes out of a tree of selects (pick64) -- visiting the 64 registers with
                            // scalar branches on the wave-wide union of the words cost 0.6 ms of a 5.8-ms launch (taken branches
                            // and SALU chains, not the atomics), and the exchange rows are not free to hold the powers: the next
                            // segment is landing there.
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
                                flush_stage(p, s, stg, stg_n);
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
                                    const uint2 cell = make_uint2(key0 + ((uint32_t)(64 * r) << p.tbits), __float_as_uint(pick64(P, r)));
                                    if (staged) {
                                        stg[idx++] = cell;
                                    } else {
                                        const uint32_t slot = atomicAdd(&p.hot_count[s * kBuckets + bkt], 1u);
                                        if (slot < (uint32_t)p.hot_cap) p.hot[((int64_t)s * kBuckets + bkt) * p.hot_cap + slot] = cell;
                                    }
                                }
                                if (staged) stg_n += cells;
                            }
                        }
                    }
                    next_hot = hot;
                }

                if constexpr (RT_W64_Q2AHEAD != 0 && MODE != 3) {
                    // the step's powers are consumed: quarter 2 of the next step's segment into sixteen registers (requested right behind the powers --
                    // ahead of the stores, the threshold test and the emission -- the sixteen pairs stayed live across that block and the
                    // kernel spilled 64 registers inside the step)
                    if (nxt >= 0) {
                        __builtin_amdgcn_sched_barrier(0);
                        const rsrc_t rn = make_rsrc(stream_iq + (int64_t)nxt * N, (uint32_t)(N * sizeof(raw_t)));
#pragma unroll
                        for (int j = 0; j < 16; ++j) q2r[j] = buf_load_iq(rn, lane * (int)sizeof(raw_t), 64 * (2 + 4 * j) * (int)sizeof(raw_t), raw_t{});
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }

#ifdef RT_STAMPS
                ++st_steps;
#endif
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

#ifdef RT_STAMPS
            RT_STAMP(10);  // the last step's threshold test and emission
            st_acc[11] = st_steps;
            st_acc[12] = st_prev - st_t0;
            st_acc[13] = (uint32_t)__builtin_amdgcn_s_memrealtime() - st_r0;
            st_acc[14] = st_wave_start;
            st_acc[15] = (uint32_t)__builtin_amdgcn_s_memrealtime();
            if (p.dbg && lane == 0) {
#pragma unroll
                for (int k = 0; k < kStamps; ++k) p.dbg[(int64_t)item * 4 * kStamps + k] = st_acc[k];
            }
#endif
            if constexpr (EMIT) {
                if (stg_n) flush_stage(p, s, stg, stg_n);
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
                if constexpr (LIN) {
                    if (p.dc_flag) {  // guard of the detrend by linearity (StftParams::dc_flag): the quietest bin (the bins 0 and +-1 aside), the total
                        float mn = 3.0e38f, tot = 0.f;
#pragma unroll
                        for (int r = 0; r < 64; ++r) {
                            const bool dc_bin = (r == 0 && lane <= 1) || (r == 63 && lane == 63);
                            mn = fminf(mn, dc_bin ? 3.0e38f : acc[r]);
                            tot += acc[r];
                        }
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) {
                            mn = fminf(mn, __shfl_xor(mn, o));
                            tot += __shfl_xor(tot, o);
                        }
                        if (dc_acc > p.dc_limit * mn && dc_acc > p.dc_limit2 * tot && lane == 0) p.dc_flag[s] = 1;
                    }
                }
                // one partial row of sums per item (= chunk): the detection adds a stream's rows in chunk order, float64
                const rsrc_t rp = make_rsrc(p.psum + ((int64_t)s * p.blocks_per_stream + chunk) * N, (uint32_t)(N * sizeof(float)));
#pragma unroll
                for (int r = 0; r < 64; ++r) raw_buffer_store_f1(acc[r], rp, lane * 4, 256 * r, 0);
                if (p.chunk_min) {
                    // the item's row of chunk minima (StftParams::chunk_min): this chunk's sums where it is a complete one of full length
                    const bool whole = c_len == L && (c0 + L <= T);
                    const rsrc_t rm = make_rsrc(p.chunk_min + ((int64_t)s * p.blocks_per_stream + chunk) * N, (uint32_t)(N * sizeof(float)));
#pragma unroll
                    for (int r = 0; r < 64; ++r) raw_buffer_store_f1(whole ? acc[r] : __uint_as_float(0x7f7f7f7fu), rm, lane * 4, 256 * r, 0);
                }
            }
        } while (false);
        item = n_waves + __builtin_amdgcn_readfirstlane((int)ticket);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): a prefetch nobody took has landed
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
