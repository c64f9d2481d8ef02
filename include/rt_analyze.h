/*
 * rt_analyze.h -- C-ABI of the MI355X-native signal-analysis path.
 *
 * This library replaces ONE path of Nature40/pyradiotracking: the per-buffer
 * callback SignalAnalyzer.process_samples (reference
 * radiotracking/analyze.py:192-268) -- STFT power (analyze.py:234-241, i.e.
 * scipy.signal.spectrogram), plateau extraction with look-back into the
 * previous buffer (analyze.py:330-452) and the shadow filter
 * (analyze.py:282-328) -- batched over many independent streams resident in
 * HBM.  The reference is pure Python and has no FFI of its own; these entry
 * points are what a ctypes binding inside the reference's SignalAnalyzer
 * would call (see INTEGRATION.md for that binding).
 *
 * Conventions
 *   - plain C types only; every function returns an rt_status (0 = ok, <0 =
 *     error) except where noted; no exception crosses the boundary.
 *   - one handle = one GPU + one HIP stream + the per-stream carried state
 *     (the look-back tail that replaces `_spectrogram_last`, analyze.py:268).
 *     A handle is not thread-safe.
 *   - IQ is complex64 (interleaved float32 I,Q), stream-major:
 *     sample b of stream s at iq[s * stream_stride + b].
 *   - results are integer cell coordinates plus float32 linear powers; the
 *     float64 / datetime part of a Signal (frequency, ts, duration, dB) is
 *     derived on the host from them (pyradiotracking_amd/analyze.py), so it is
 *     bit-exact by construction.
 */
#ifndef RT_ANALYZE_H
#define RT_ANALYZE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 6 (round 6): fft_nperseg 32 / 64 / 128 / 8192 / 16384 run fused scans (every mode they have); record_capacity is where a stream's
 * room starts, not a limit (rt_fetch); rt_format.h: rt_signal_rows_from_records, rt_host_set_threads; rt_match.h:
 * rt_match_add_many, rt_match_pending_count_many.  Nothing of version 5 was removed or changed in layout. */
#define RT_ABI_VERSION 6

typedef enum rt_status {
    RT_OK = 0,
    RT_E_INVALID = -1,      /* bad argument / configuration                     */
    RT_E_UNSUPPORTED = -2,  /* e.g. nperseg < 8, > 8192 and not a power of two  */
    RT_E_NO_DEVICE = -3,    /* no usable GPU / HIP failure at create            */
    RT_E_HIP = -4,          /* HIP runtime error (see rt_last_error)            */
    RT_E_CAPACITY = -5,     /* record capacity exceeded (results truncated, never dropped: rt_fetch) */
    RT_E_ONE_SEGMENT = -6,  /* exactly one segment: the reference raises
                               IndexError there (analyze.py:354, times[1])      */
    RT_E_NOMEM = -7,
    RT_E_HOT_OVERFLOW = -8  /* RT_MODE_SPARSE only: a candidate list overflowed (hot_capacity); the
                               call produced NO result and has been dropped -- unlike RT_E_CAPACITY,
                               which hands out a truncated result                                  */
} rt_status;

/* how the batch is analysed */
typedef enum rt_mode {
    RT_MODE_AUTO = 0,   /* fused sparse path; a buffer whose candidate lists overflow is re-run one
                           level up (RT_MODE_PREFILTER, RT_MODE_RUNFILTER where available, then dense) and the handle stays
                           on that level for the next 16 buffers (32, 64 ... 1024 while the probes of
                           the level below keep overflowing) */
    RT_MODE_DENSE = 1,  /* materialise the power spectrogram (any input)        */
    RT_MODE_SPARSE = 2, /* fused sparse path only; overflow -> RT_E_HOT_OVERFLOW */
    RT_MODE_PREFILTER = 3, /* sparse path behind the run-length pre-filter (two scan passes: per chunk of
                             segments and bin "every cell passes the absolute threshold", then candidate
                             cells only from such chunks and their neighbours): for inputs whose noise
                             crosses the threshold.  Needs signal_min_duration >= 2 * segs_per_chunk hops
                             (else RT_E_UNSUPPORTED); overflow -> RT_E_HOT_OVERFLOW.  RT_MODE_AUTO goes
                             through it between the sparse and the dense path where it is available. */
    RT_MODE_RUNFILTER = 4 /* ABI v5: sparse path behind the EXACT run-length pre-filter: a first scan keeps the
                             threshold bit of every cell, a planning kernel keeps the cells of threshold runs of
                             at least the minimum plateau length (or through t = 0), a second scan transforms only
                             the segments that hold such cells.  The bits also ask for snr_threshold x the bin's
                             quiet level (from the buffer before, verified against this buffer's row means), so the
                             level stays selective with the noise floor over the absolute threshold.  Any
                             segs_per_chunk.  RT_MODE_AUTO uses it between the sparse (or RT_MODE_PREFILTER, where
                             that exists) and the dense path -- at the reference's defaults (300 kS/s, 8 ms) it is
                             the only level in between.  RT_E_UNSUPPORTED where the minimum plateau length does not
                             fit the planning tiles; overflow -> RT_E_HOT_OVERFLOW. */
} rt_mode;

/*
 * Analyzer configuration.  Mirrors the derived parameters of
 * SignalAnalyzer.__init__ (analyze.py:101-117) plus batch geometry.
 */
typedef struct rt_config {
    int32_t device;             /* HIP device ordinal                                    */
    int32_t n_streams;          /* S: independent streams analysed per call              */
    int32_t nperseg;            /* fft_nperseg (analyze.py:111; the reference passes any integer on to SciPy, __main__.py:59):
                                   any size from 8 to 8192, or a power of two up to 16384.  The powers of two 256 ... 4096 run the
                                   fused scan kernels (every rt_mode); every other size runs a general transform on the dense path
                                   (other powers of two: radix-2 in LDS; the rest: Bluestein's algorithm on it) -- RT_MODE_AUTO or
                                   RT_MODE_DENSE only, 16 bytes of traffic per sample; anything else: RT_E_UNSUPPORTED */
    int32_t mode;               /* rt_mode                                               */
    int64_t max_samples;        /* largest per-stream buffer length B accepted           */
    double sample_rate;         /* fs (analyze.py:101)                                   */
    const float *window;        /* host pointer, nperseg float32 coefficients: the
                                   window cast to the IQ dtype as SciPy does
                                   (scipy/signal/_spectral_py.py:2083-2084)              */
    float scale;                /* 1/(fs*sum(w*w)) in float32 (_spectral_py.py:2087)     */
    float threshold;            /* signal_threshold, linear (analyze.py:115)             */
    float snr_threshold;        /* snr_threshold, linear (analyze.py:116)                */
    float calibration_db;       /* only used to order maxima in the shadow filter        */
    double min_duration_s;      /* signal_min_duration (analyze.py:113)                  */
    double max_duration_s;      /* signal_max_duration (analyze.py:114)                  */
    int32_t hot_capacity;       /* sparse path: candidate cells kept per (stream, bin mod 16 bucket) and call
                                   (0 = default: one full bin row times max(1, nperseg / 1024), 1024..8192)      */
    int32_t record_capacity;    /* records per stream and call the handle has room for AT FIRST (0 = default 1024): a stream
                                   that finds more grows the capacity -- the call is analysed again inside rt_fetch --, as
                                   the reference appends without limit (analyze.py:449-450).  Only rt_extract truncates. */
    int32_t segs_per_chunk;     /* segments per lane-group chunk (0 = default)           */
    int32_t flags;              /* RT_FLAG_*                                             */
    void *hip_stream;           /* hipStream_t to launch on, or NULL for an own stream   */
    int32_t lanes;              /* 0 / 1 = one launch sequence per call.  n > 1: the streams are split into n
                                   contiguous groups, each analysed on its own HIP stream (hip_stream must be
                                   NULL), so that the detection kernels and launch gaps of one group overlap the
                                   scan of another; same records, rt_fetch still returns them in stream order     */
    int32_t record_pool;        /* records the pinned result pool of a call holds at first (0 = default:
                                   min(n_streams * record_capacity, 4 Mi)).  A call that needs more grows the pool
                                   and is analysed again when it is fetched, so nothing is lost up to
                                   record_capacity records per stream (the reference appends without limit,
                                   analyze.py:449-450)                                                            */
} rt_config;

#define RT_FLAG_TIMING 1u /* record HIP events around the kernels of each call */
#define RT_FLAG_NO_LIN_DETREND 2u /* always subtract the segment mean before windowing (scipy's order of operations);
                                     default: for hamming / hann / boxcar windows and complex64 input the constant
                                     detrend is applied to the transform instead (three bins), which is cheaper and
                                     equal within float32 round-off -- other windows and uint8 input (where a saturated
                                     segment cancels exactly in the reference) use the subtract-first form anyway */
#define RT_FLAG_GROUP_DETECT 4u    /* sparse detection (analyze.py:330-452 on the candidate lists) with one wave per stream, or per quarter
                                     of a stream's sixteen lists, instead of one per list -- the same records; the default from 1 024
                                     streams per handle on, while the streams of the call fetched last held few candidate cells (<= 448 on
                                     average).  This flag: at any number of streams, whatever they hold.  Only where the fused kernels have
                                     the form (nperseg <= 256); ignored elsewhere */
#define RT_FLAG_NO_GROUP_DETECT 8u /* ... never */

/*
 * One extracted plateau, before it becomes a Signal (analyze.py:442-449).
 * `start` may be negative: it then indexes the previous buffer from its end,
 * exactly like the reference's negative `start` (analyze.py:383-388, 422-423).
 */
typedef struct rt_record {
    int32_t stream;   /* stream index within the batch                                   */
    int32_t fi;       /* frequency bin, fftfreq order (analyze.py:357)                   */
    int32_t start;    /* first cell of `data` (analyze.py:437-440)                       */
    int32_t end;      /* one past the last cell                                          */
    float max_p;      /* max(data), linear                                               */
    float mean_p;     /* mean(data), linear                                              */
    float std_db;     /* std(10*log10(data)), population                                 */
    float row_mean;   /* mean of the bin's row over the whole buffer (`freq_avg`, :375)  */
    int32_t shadowed; /* 1 if filter_shadow_signals drops it (analyze.py:315-328)        */
    int32_t reserved;
} rt_record;

typedef struct rt_handle rt_handle;

int rt_abi_version(void);

/* Create an analyzer for `cfg` on cfg->device.  Allocates all device scratch. */
int rt_create(const rt_config *cfg, rt_handle **out);

void rt_destroy(rt_handle *h);

/* Forget the carried look-back state (== `_spectrogram_last = None`, analyze.py:128). */
int rt_reset(rt_handle *h);

/*
 * Forget the look-back state of ONE stream: what the reference's Runner does when it replaces a dead or
 * timed-out SDR's analyzer by a new one (__main__.py:153-190 -> a fresh SignalAnalyzer, analyze.py:128).
 * Takes effect with the next rt_process; the other streams keep their state.  [SURVEY 8(f) rank 4]
 */
int rt_reset_stream(rt_handle *h, int32_t stream);

/*
 * Per-stream thresholds: the reference runs one SignalAnalyzer per SDR, each with its own
 * `calibration_db` (__main__.py:140-141: zip(device, calibration)), and the absolute threshold depends
 * on it (analyze.py:115: from_dB(signal_threshold_dbw + calibration_db)).  `threshold` and
 * `calibration_db` are HOST arrays of n_streams float32 (linear threshold; calibration in dB, used as in
 * rt_config to order maxima in the shadow filter); either may be NULL = keep rt_config's value for every
 * stream.  Applies to calls enqueued afterwards; refused (RT_E_INVALID) while unfetched calls are pending, since
 * AUTO mode may still re-run those with the thresholds they were enqueued with.  A stream whose threshold CHANGES
 * starts its next buffer without look-back, as after rt_reset_stream: in the reference a threshold is fixed
 * when the SignalAnalyzer is built (analyze.py:115), so a new one means a new analyzer.  [SURVEY 8(f) rank 4]
 */
int rt_set_stream_params(rt_handle *h, const float *threshold, const float *calibration_db);

/*
 * Analyse one buffer per stream: the body of process_samples (analyze.py:234-251,
 * 268).  `iq_dev` is a DEVICE pointer to S*stream_stride complex64; n_samples =
 * len(buffer) (<= max_samples); stream_stride in samples (>= n_samples).
 * Asynchronous: enqueues on the handle's streams.  Results via rt_fetch.
 * Up to two calls may be in flight (enqueue call k+1 before fetching call k, so the
 * GPU never waits for the host; with cfg.lanes > 1 the lanes' kernels also overlap
 * each other); a third rt_process without
 * an rt_fetch drops the oldest unfetched result.  `iq_dev` must stay valid and
 * unchanged until the call has been fetched.  `iq_dev` must be 8-byte aligned (whole complex64
 * samples; rt_process_u8: 2-byte aligned) -- anything else is refused with RT_E_INVALID, not launched.
 */
int rt_process(rt_handle *h, const void *iq_dev, int64_t n_samples, int64_t stream_stride);

/*
 * Same for the RTL-SDR wire format: `iq_u8_dev` is a DEVICE pointer to S*stream_stride samples
 * of interleaved uint8 (I, Q) -- 2 bytes per sample, what librtlsdr delivers before pyrtlsdr's
 * packed_bytes_to_iq (the producer of the buffer handed to process_samples, analyze.py:157).
 * The conversion (byte/127.5 - 1) is fused into the scan kernel's load (one float32 fma per
 * component, <= 1 float32 ulp from pyrtlsdr's float64 expression); everything after it is the
 * complex64 path.  n_samples / stream_stride count samples, not bytes.  [SURVEY 8(f) rank 1]
 */
int rt_process_u8(rt_handle *h, const void *iq_u8_dev, int64_t n_samples, int64_t stream_stride);

/*
 * Same with IQ in host memory: copied (blocking) to an internal device buffer first -- one per call in flight, so
 * the caller may reuse its buffer as soon as the call returns and a call's samples stay in place until it is
 * fetched.  rt_process_u8_host takes what librtlsdr's read callback delivers (interleaved uint8 I,Q in host
 * memory): the direct replacement of `sdr.read_samples_async(self.process_samples, ...)` + packed_bytes_to_iq
 * (analyze.py:157) for a binding that registers a bytes callback instead.
 */
int rt_process_host(rt_handle *h, const void *iq_host, int64_t n_samples, int64_t stream_stride);
int rt_process_u8_host(rt_handle *h, const void *iq_u8_host, int64_t n_samples, int64_t stream_stride);

/*
 * Wait for the OLDEST unfetched rt_process / rt_extract and copy its records, ordered by
 * (stream, fi, start) -- the reference's emission order per stream
 * (analyze.py:357, 364).  Records carry the shadow verdict; none is removed.
 * *n_out receives the number of records available; at most `cap` are written.
 * With out == NULL (or cap == 0) and records available the call is only a size
 * query: the result stays pending until it is fetched with a buffer.  A fetch
 * with a buffer consumes the call whatever `cap` is (records beyond `cap` are
 * lost; with cfg.lanes > 1 in every lane alike).
 * RT_E_CAPACITY: the result is truncated (and still delivered).  Neither the record
 * pool of a call (ABI v5) nor the per-stream record capacity (round 6) is a limit for
 * rt_process*: a call that finds more records than either holds grows it and is
 * analysed again inside this function -- only rt_extract (whose spectrogram the
 * library does not keep) or a device / host without memory for the larger areas end
 * in RT_E_CAPACITY, and then every stream still delivers the first records, in (bin,
 * start) order -- the reference's append order -- that fit (never an empty list).  RT_E_HOT_OVERFLOW (RT_MODE_SPARSE): no
 * result, the call is consumed.
 * If an rt_process fails, nothing stays enqueued for it (with lanes: in no lane),
 * and the look-back state is the one before the call.
 */
int rt_fetch(rt_handle *h, rt_record *out, size_t cap, size_t *n_out);

/*
 * extract_signals + filter_shadow_signals on a caller-supplied power
 * spectrogram (analyze.py:330-452 with explicit arguments): `spec_dev` is a
 * DEVICE pointer to [S][n_seg][n_bins] float32 (segment-major, the memory
 * layout SciPy's result has under its [F,T] view).  `last_dev` is the previous
 * spectrogram in the same layout with n_seg_last segments, or NULL
 * (`_spectrogram_last is None`).  n_bins is free (not tied to nperseg).
 * Does not touch the carried state.  Results via rt_fetch.
 */
int rt_extract(rt_handle *h, const float *spec_dev, int32_t n_seg, int32_t n_bins,
               const float *last_dev, int32_t n_seg_last);

/*
 * Debug / test entry: STFT power only.  Writes [S][T][nperseg] float32 to
 * `spec_dev` (device), T = n_samples / nperseg.  Synchronous.
 */
int rt_spectrogram(rt_handle *h, const void *iq_dev, int64_t n_samples, int64_t stream_stride,
                   float *spec_dev);

/*
 * Profiling aid: launches the scan kernel's load stream only (same grid, same
 * addresses, same prefetch; no arithmetic, no stores).  Its byte count is
 * known exactly -- S * (T + one halo segment per chunk) * nperseg * 8 -- so a
 * rocprofv3 --pmc FETCH_SIZE pass over it calibrates the counter for this
 * access shape (8-byte loads; MI355X_MICROARCH.md "HBM").  Synchronous.
 */
int rt_calibrate_read(rt_handle *h, const void *iq_dev, int64_t n_samples, int64_t stream_stride);

/* Per-call figures of the last rt_process (valid after rt_fetch). */
typedef struct rt_call_info {
    int32_t n_seg;            /* T of the call                                           */
    int32_t mode_used;        /* RT_MODE_DENSE, RT_MODE_SPARSE, RT_MODE_PREFILTER or RT_MODE_RUNFILTER */
    int32_t fell_back;        /* 1 if candidate lists overflowed and (part of) the call was re-run */
    int32_t n_dense_streams;  /* RT_MODE_AUTO: streams re-run dense on their own because only they overflowed
                                 (a few noisy SDRs in a batch; mode_used then still names the batch's path) */
    int64_t n_hot;            /* candidate cells emitted by the sparse scan              */
    int64_t n_records;        /* records produced                                        */
    float ms_stft;            /* RT_FLAG_TIMING: STFT/scan kernel, HIP events, ms        */
    float ms_detect;          /* RT_FLAG_TIMING: detect kernel(s), ms                    */
    float ms_total;           /* RT_FLAG_TIMING: first launch to last launch, ms         */
    int32_t segs_per_chunk;   /* the handle's chunk length (rt_config.segs_per_chunk, or what 0 chose); was reserved */
} rt_call_info;

int rt_get_call_info(rt_handle *h, rt_call_info *info);

/* Message of the last error on this handle (or of the last failed rt_create if h == NULL). */
const char *rt_last_error(rt_handle *h);

/* Plain device-memory helpers so that a host without its own HIP binding
 * (ctypes-only integration) can stage IQ: thin hipMalloc/hipFree/hipMemcpy. */
int rt_dev_alloc(int32_t device, size_t bytes, void **out);
int rt_dev_free(int32_t device, void *ptr);
int rt_dev_upload(int32_t device, void *dst_dev, const void *src_host, size_t bytes);
int rt_dev_download(int32_t device, void *dst_host, const void *src_dev, size_t bytes);
int rt_device_count(int *count);

#ifdef __cplusplus
}
#endif
#endif /* RT_ANALYZE_H */
