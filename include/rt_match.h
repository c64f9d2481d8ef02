/*
 * rt_match.h -- C-ABI of the cross-SDR signal matcher.
 *
 * Replaces, for batches of records, the reference's SignalMatcher.add
 * (radiotracking/match.py:54-82) together with the MatchingSignal arithmetic it
 * drives (radiotracking/__init__.py:279-406): signals of one station's SDRs
 * are grouped when frequency, time and (optionally) duration agree within the
 * configured tolerances; a group is handed on once a later signal arrives more
 * than the timeout after it.  The algorithm is sequential and order dependent
 * (greedy first match in list order), works on a handful of open groups, and
 * is pure host code: no GPU is involved and none is needed to call it.
 *
 * All times are integer microseconds, the resolution of the datetime /
 * timedelta values the reference compares (timestamps: microseconds since the
 * Unix epoch, UTC), so every comparison is exact; frequencies and powers are
 * float64 like the reference's Python floats.
 */
#ifndef RT_MATCH_H
#define RT_MATCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* SignalMatcher.__init__ (match.py:33-50).  The three float parameters are
 * turned into whole microseconds the way datetime.timedelta(seconds=..) /
 * (milliseconds=..) does (round half to even). */
typedef struct rt_match_config {
    int32_t n_devices;        /* len(device): columns of the per-device power list              */
    int32_t reserved;
    double timeout_s;         /* matching_timeout_s                                              */
    double time_diff_s;       /* matching_time_diff_s                                            */
    double bandwidth_hz;      /* matching_bandwidth_hz                                           */
    double duration_diff_ms;  /* matching_duration_diff_ms; 0 or NaN = None (no duration check)  */
} rt_match_config;

/* The fields of a Signal the matcher reads (__init__.py:110-170). */
typedef struct rt_match_signal {
    int32_t device;      /* index into the device list; any other value is a device without a
                            column (it still takes part in ts / frequency / duration)            */
    int32_t reserved;
    int64_t ts_us;       /* Signal.ts                                                            */
    int64_t duration_us; /* Signal.duration                                                      */
    double frequency;    /* Signal.frequency                                                     */
    double avg;          /* Signal.avg                                                           */
} rt_match_signal;

/* A MatchingSignal as its consumers see it (__init__.py:296-340). */
typedef struct rt_matched {
    int64_t ts_us;       /* min over members          (:307-316)                                 */
    int64_t duration_us; /* max over members          (:296-305)                                 */
    double frequency;    /* statistics.median         (:318-327)                                 */
    int32_t n_members;
    int32_t reserved;
} rt_matched;

typedef struct rt_matcher rt_matcher;

int rt_match_create(const rt_match_config *cfg, rt_matcher **out);
void rt_match_destroy(rt_matcher *m);

/* Forget all open groups. */
int rt_match_reset(rt_matcher *m);

/* Number of open (not yet consumed) groups. */
int rt_match_pending_count(rt_matcher *m, size_t *n_out);

/*
 * SignalMatcher.add for `n` signals in order (match.py:54-82).  Groups that time
 * out on the way are appended to `out` in the order the reference would put
 * them on its queue.  `out_avgs` receives n_devices doubles per group (the
 * `_avgs` list, __init__.py:329-339) and `out_present` n_devices bytes per
 * group (1 = that device has a member, 0 = the reference's None; the double is
 * NaN there); either may be NULL.  At most pending + n groups can come out:
 * with `cap` smaller than that the call fails with RT_E_CAPACITY before it
 * changes anything.
 */
int rt_match_add(rt_matcher *m, const rt_match_signal *sigs, size_t n, rt_matched *out, double *out_avgs,
                 uint8_t *out_present, size_t cap, size_t *n_out);

/*
 * rt_match_add for MANY matchers in one call -- one matcher per station (the reference runs one SignalMatcher per station
 * process, match.py:21-50; its rule is sequential inside a station and independent between stations) --, the matchers dealt to
 * `rt_host_set_threads` threads (include/rt_format.h).  Matcher k takes the signals sigs[sig_offsets[k] .. sig_offsets[k + 1]) in
 * order and writes its timed-out groups from out[out_offsets[k]] on (out_avgs / out_present: rows of ITS n_devices columns from
 * element out_offsets[k] * n_devices_max on, n_devices_max = the largest n_devices of the set); out_offsets[k + 1] - out_offsets[k]
 * must be at least that matcher's pending groups + signals.  n_out[k] receives its number of groups.  The result is what n_matchers
 * calls of rt_match_add produce.  Returns the first error (and leaves the matchers after the failing one's thread block untouched
 * only on RT_E_INVALID / RT_E_CAPACITY, which are checked for every matcher before any is changed).
 */
/* rt_match_pending_count of every matcher of a set in one call (sizes the output of rt_match_add_many). */
int rt_match_pending_count_many(rt_matcher *const *ms, size_t n_matchers, size_t *n_out);

int rt_match_add_many(rt_matcher *const *ms, size_t n_matchers, const rt_match_signal *sigs, const size_t *sig_offsets,
                      rt_matched *out, double *out_avgs, uint8_t *out_present, const size_t *out_offsets, int32_t n_devices_max,
                      size_t *n_out);

/* Copy of the open groups in list order (`_matched`), nothing is consumed. */
int rt_match_pending(rt_matcher *m, rt_matched *out, double *out_avgs, uint8_t *out_present, size_t cap,
                     size_t *n_out);

/* MatchingSignal.has_member of open group `index` for one signal (__init__.py:341-392): 1 / 0, <0 on error. */
int rt_match_has_member(rt_matcher *m, size_t index, const rt_match_signal *sig);

const char *rt_match_last_error(rt_matcher *m);

#ifdef __cplusplus
}
#endif
#endif /* RT_MATCH_H */
