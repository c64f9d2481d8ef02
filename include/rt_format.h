/*
 * rt_format.h -- C-ABI of the record serialisers.
 *
 * The wire / on-disk formats the reference produces for a Signal and for a
 * MatchingSignal (radiotracking/consume.py): the `;`-separated CSV row of
 * CSVConsumer.add (consume.py:192-196) and MQTTConsumer.add (:148-151; csv
 * "excel" dialect, values through csvify :50-55), the JSON document of
 * MQTTConsumer.add (:141-145, json.dumps(as_dict, default=jsonify), jsonify
 * :23-32) and its CBOR message (:154-160, cbor2.dumps(as_list, timezone=utc,
 * datetime_as_timestamp=True, default=cborify), cborify :35-39: timedelta ->
 * tag 1337 around the seconds).
 * Here they are produced for whole arrays of records at once; host code, no
 * GPU involved.
 *
 * Value formatting follows the Python objects the reference serialises:
 *   - a float is written as repr(float) (shortest digits that round-trip,
 *     fixed notation for 1e-4 <= |x| < 1e16, else d.ddde+XX; json: NaN,
 *     Infinity; csv: nan, inf),
 *   - Time is an aware UTC datetime: str() "YYYY-MM-DD HH:MM:SS[.ffffff]+00:00"
 *     in CSV, isoformat() (with "T") in JSON, epoch seconds in CBOR (tag 1,
 *     integer when the microsecond field is zero, else float64),
 *   - Duration is timedelta.total_seconds() = microseconds / 1e6,
 *   - a missing per-device power (None) is an empty CSV field, JSON null,
 *     CBOR null.
 *
 * Every function writes `n` messages back to back into `out` and their start
 * offsets into `offsets[0..n]` (offsets[n] = total bytes); CSV rows end with
 * "\r\n" (the csv module's line terminator), JSON / CBOR messages have no
 * separator.  If `cap` is too small nothing useful is written, *n_bytes
 * receives the size needed and RT_E_CAPACITY is returned (call with cap = 0
 * to size the buffer).
 */
#ifndef RT_FORMAT_H
#define RT_FORMAT_H

#include <stddef.h>
#include <stdint.h>

#include "rt_analyze.h" /* rt_record */

#ifdef __cplusplus
extern "C" {
#endif

/* The nine fields of a Signal (radiotracking/__init__.py:136-196) as numbers. */
typedef struct rt_signal_row {
    int32_t device;      /* index into `device_names`                                   */
    int32_t reserved;
    int64_t ts_us;       /* Signal.ts, microseconds since the Unix epoch (UTC)           */
    int64_t duration_us; /* Signal.duration                                              */
    double frequency;
    double max_dbw, avg_dbw, std_db, noise_dbw, snr_db;
} rt_signal_row;

/* The fixed part of a MatchedSignal (radiotracking/__init__.py:223-268); its per-device
 * powers travel as `avgs` [n][n_devices] doubles + `present` [n][n_devices] bytes
 * (0 = None), the layout rt_match_add produces. */
typedef struct rt_matched_row {
    int64_t ts_us;
    int64_t duration_us;
    double frequency;
} rt_matched_row;

typedef enum rt_format_kind {
    RT_FORMAT_CSV = 0,  /* consume.py:148-151, 192-196 */
    RT_FORMAT_JSON = 1, /* consume.py:141-145          */
    RT_FORMAT_CBOR = 2  /* consume.py:154-160          */
} rt_format_kind;

/* Signal messages.  `device_names`: n_devices NUL-terminated UTF-8 strings (Signal.device). */
int rt_format_signals(int32_t kind, const rt_signal_row *rows, size_t n, const char *const *device_names,
                      int32_t n_devices, char *out, size_t cap, size_t *offsets, size_t *n_bytes);

/* MatchedSignal / MatchingSignal messages.  `device_names` are the column names (header :252-259). */
int rt_format_matched(int32_t kind, const rt_matched_row *rows, const double *avgs, const uint8_t *present, size_t n,
                      const char *const *device_names, int32_t n_devices, char *out, size_t cap, size_t *offsets,
                      size_t *n_bytes);

/* repr(float) of one value into buf (>= 32 bytes); returns the length.  Exposed for tests. */
int rt_format_float_repr(double x, char *buf);

/*
 * Signal rows of the records of one analysis call, for whole arrays (the conversion the reference does per signal at
 * analyze.py:420-449, __init__.py:110-170): start time and duration from the cell coordinates through the reference's float64
 * expressions and datetime.timedelta's rounding, on `rt_host_set_threads` threads.  `rec`: n records (the caller drops shadowed
 * ones first: the reference never hands them to a consumer, analyze.py:248-251); `ts_start_us[stream]`: the buffer's start
 * (microseconds since the epoch, UTC); `frequency` and the five float32 dB columns: per record, evaluated by the caller with the
 * reference's own NumPy expressions (their last digit is printed by the CSV / JSON consumers).
 */
int rt_signal_rows_from_records(const rt_record *rec, size_t n, int32_t nperseg, double sample_rate, const int64_t *ts_start_us,
                                int32_t n_streams, const double *frequency, const float *max_dbw, const float *avg_dbw, const float *std_db,
                                const float *noise_dbw, const float *snr_db, rt_signal_row *out);

/*
 * The records a consumer sees -- those with `shadowed == 0` (analyze.py:248-251: the reference hands only the filtered list on), in
 * their order -- copied to `out` (room for n records); *n_kept = their number.  On `rt_host_set_threads` threads.
 */
int rt_records_keep_unshadowed(const rt_record *rec, size_t n, rt_record *out, size_t *n_kept);

/*
 * Threads the host-side sinks (rt_format_*, rt_signal_rows_from_records, rt_records_keep_unshadowed, rt_match_add_many) use per call: n > 0 exactly n, 0 =
 * automatic (the machine's hardware threads, at most 32).  Process-wide; returns the number in force.  The output of every sink is
 * byte for byte the same for any number of threads (blocks of rows / whole matchers per thread, assembled in order).
 */
int rt_host_set_threads(int32_t n);

#ifdef __cplusplus
}
#endif
#endif /* RT_FORMAT_H */
