// rt_format.cpp -- host implementation of include/rt_format.h: the reference's CSV / JSON / CBOR
// messages (radiotracking/consume.py:23-55, 141-160, 192-196) for arrays of records.
//
// What is restated here is the behaviour of the Python standard library (csv "excel" dialect with
// delimiter ';', json.dumps with its defaults, repr(float), str()/isoformat() of an aware UTC
// datetime) and of cbor2's encoder (RFC 8949 definite-length items; floats always as float64 except
// NaN / infinities as float16; datetime_as_timestamp -> tag 1) on the values the reference hands them.
#include <hip/hip_runtime.h>  // hipcc compiles this file as HIP too

#include <algorithm>
#include <atomic>
#include <charconv>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_analyze.h"
#include "../../include/rt_format.h"
#include "rt_core.h"
#include "rt_hostpar.h"

namespace {

// ---- repr(float) ---------------------------------------------------------------------------------
// CPython float_repr_style 'short', format code 'r': shortest round-trip digits; exponent notation
// iff decpt <= -4 or decpt > 16 (value = 0.DIGITS x 10^decpt); ".0" appended to integral values.
int float_repr(double x, char *out) {
    if (std::isnan(x)) {
        std::memcpy(out, "nan", 3);
        return 3;
    }
    if (std::isinf(x)) {
        const char *s = x > 0 ? "inf" : "-inf";
        const int n = (int)std::strlen(s);
        std::memcpy(out, s, n);
        return n;
    }
    char *p = out;
    if (std::signbit(x)) {
        *p++ = '-';
        x = -x;
    }
    if (x == 0.0) {
        std::memcpy(p, "0.0", 3);
        return (int)(p - out) + 3;
    }
    char sci[40];
    const auto r = std::to_chars(sci, sci + sizeof sci, x, std::chars_format::scientific);  // d[.ddd]e[+-]XX
    char digits[24];
    int nd = 0, e = 0;
    const char *q = sci;
    for (; q < r.ptr && *q != 'e'; ++q)
        if (*q != '.') digits[nd++] = *q;
    ++q;  // 'e'
    const bool eneg = (*q == '-');
    ++q;
    for (; q < r.ptr; ++q) e = e * 10 + (*q - '0');
    if (eneg) e = -e;
    const int decpt = e + 1;
    if (decpt <= -4 || decpt > 16) {
        *p++ = digits[0];
        if (nd > 1) {
            *p++ = '.';
            std::memcpy(p, digits + 1, nd - 1);
            p += nd - 1;
        }
        *p++ = 'e';
        int ex = decpt - 1;
        *p++ = ex < 0 ? '-' : '+';
        if (ex < 0) ex = -ex;
        char eb[8];
        int ne = 0;
        do {
            eb[ne++] = (char)('0' + ex % 10);
            ex /= 10;
        } while (ex);
        if (ne < 2) eb[ne++] = '0';
        while (ne) *p++ = eb[--ne];
    } else if (decpt <= 0) {
        *p++ = '0';
        *p++ = '.';
        for (int i = 0; i < -decpt; ++i) *p++ = '0';
        std::memcpy(p, digits, nd);
        p += nd;
    } else if (decpt >= nd) {
        std::memcpy(p, digits, nd);
        p += nd;
        for (int i = 0; i < decpt - nd; ++i) *p++ = '0';
        *p++ = '.';
        *p++ = '0';
    } else {
        std::memcpy(p, digits, decpt);
        p += decpt;
        *p++ = '.';
        std::memcpy(p, digits + decpt, nd - decpt);
        p += nd - decpt;
    }
    return (int)(p - out);
}

void put_float_py(std::string &o, double x) {  // csv: repr(float)
    char b[40];
    o.append(b, (size_t)float_repr(x, b));
}

void put_float_json(std::string &o, double x) {  // json.encoder.floatstr
    if (std::isnan(x)) {
        o += "NaN";
    } else if (std::isinf(x)) {
        o += x > 0 ? "Infinity" : "-Infinity";
    } else {
        put_float_py(o, x);
    }
}

// ---- datetime --------------------------------------------------------------------------------------
struct Civil {
    int64_t y;
    int m, d, hh, mm, ss, us;
};

Civil civil_from_us(int64_t ts_us) {
    int64_t days = ts_us / 86400000000LL, rem = ts_us % 86400000000LL;
    if (rem < 0) {
        rem += 86400000000LL;
        days -= 1;
    }
    // days since 1970-01-01 -> proleptic Gregorian date (era arithmetic on 400-year cycles)
    const int64_t z = days + 719468;
    const int64_t era = (z >= 0 ? z : z - 146096) / 146097;
    const int64_t doe = z - era * 146097;
    const int64_t yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
    const int64_t doy = doe - (365 * yoe + yoe / 4 - yoe / 100);
    const int64_t mp = (5 * doy + 2) / 153;
    Civil c;
    c.d = (int)(doy - (153 * mp + 2) / 5 + 1);
    c.m = (int)(mp < 10 ? mp + 3 : mp - 9);
    c.y = yoe + era * 400 + (c.m <= 2 ? 1 : 0);
    c.us = (int)(rem % 1000000);
    const int64_t secs = rem / 1000000;
    c.hh = (int)(secs / 3600);
    c.mm = (int)(secs / 60 % 60);
    c.ss = (int)(secs % 60);
    return c;
}

void put_padded(std::string &o, int64_t v, int width) {
    char b[24];
    int n = 0;
    do {
        b[n++] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    while (n < width) b[n++] = '0';
    while (n) o.push_back(b[--n]);
}

// datetime.isoformat(sep) of an aware UTC datetime: microseconds only when non-zero, offset "+00:00"
void put_datetime(std::string &o, int64_t ts_us, char sep) {
    const Civil c = civil_from_us(ts_us);
    put_padded(o, c.y, 4);
    o.push_back('-');
    put_padded(o, c.m, 2);
    o.push_back('-');
    put_padded(o, c.d, 2);
    o.push_back(sep);
    put_padded(o, c.hh, 2);
    o.push_back(':');
    put_padded(o, c.mm, 2);
    o.push_back(':');
    put_padded(o, c.ss, 2);
    if (c.us) {
        o.push_back('.');
        put_padded(o, c.us, 6);
    }
    o += "+00:00";
}

// timedelta.total_seconds(): integer microseconds / 10**6, correctly rounded
double total_seconds(int64_t us) { return (double)us / 1e6; }

// ---- csv ---------------------------------------------------------------------------------------------
// csv.writer, dialect "excel" (QUOTE_MINIMAL, doublequote) with delimiter ';': a field is quoted when it
// holds the delimiter, the quote character or a character of the line terminator "\r\n"
void put_csv_text(std::string &o, const char *s) {
    bool quote = false;
    for (const char *p = s; *p; ++p)
        if (*p == ';' || *p == '"' || *p == '\r' || *p == '\n') quote = true;
    if (!quote) {
        o += s;
        return;
    }
    o.push_back('"');
    for (const char *p = s; *p; ++p) {
        if (*p == '"') o.push_back('"');
        o.push_back(*p);
    }
    o.push_back('"');
}

// ---- json --------------------------------------------------------------------------------------------
// json.dumps defaults: ensure_ascii=True (py_encode_basestring_ascii), separators (", ", ": ")
void put_json_text(std::string &o, const char *s) {
    static const char *hex = "0123456789abcdef";
    auto u16 = [&](unsigned v) {
        o += "\\u";
        o.push_back(hex[(v >> 12) & 15]);
        o.push_back(hex[(v >> 8) & 15]);
        o.push_back(hex[(v >> 4) & 15]);
        o.push_back(hex[v & 15]);
    };
    o.push_back('"');
    const unsigned char *p = reinterpret_cast<const unsigned char *>(s);
    while (*p) {
        unsigned c = *p;
        if (c < 0x80) {
            ++p;
            switch (c) {
                case '"': o += "\\\""; break;
                case '\\': o += "\\\\"; break;
                case '\n': o += "\\n"; break;
                case '\r': o += "\\r"; break;
                case '\t': o += "\\t"; break;
                case '\b': o += "\\b"; break;
                case '\f': o += "\\f"; break;
                default:
                    // CPython escapes everything outside ' '..'~' (json.encoder ESCAPE_ASCII: [^\ -~]), DEL included
                    if (c < 0x20 || c == 0x7F) u16(c); else o.push_back((char)c);
            }
            continue;
        }
        // UTF-8 -> code point (input is trusted to be valid UTF-8; a stray byte is passed through as U+FFFD)
        unsigned cp = 0xFFFD;
        int len = 1;
        if ((c & 0xE0) == 0xC0 && (p[1] & 0xC0) == 0x80) {
            cp = ((c & 0x1F) << 6) | (p[1] & 0x3F);
            len = 2;
        } else if ((c & 0xF0) == 0xE0 && (p[1] & 0xC0) == 0x80 && (p[2] & 0xC0) == 0x80) {
            cp = ((c & 0x0F) << 12) | ((p[1] & 0x3F) << 6) | (p[2] & 0x3F);
            len = 3;
        } else if ((c & 0xF8) == 0xF0 && (p[1] & 0xC0) == 0x80 && (p[2] & 0xC0) == 0x80 && (p[3] & 0xC0) == 0x80) {
            cp = ((c & 0x07) << 18) | ((p[1] & 0x3F) << 12) | ((p[2] & 0x3F) << 6) | (p[3] & 0x3F);
            len = 4;
        }
        p += len;
        if (cp >= 0x10000) {
            const unsigned v = cp - 0x10000;
            u16(0xD800 | (v >> 10));
            u16(0xDC00 | (v & 0x3FF));
        } else {
            u16(cp);
        }
    }
    o.push_back('"');
}

// ---- cbor (RFC 8949) ------------------------------------------------------------------------------------
void put_cbor_head(std::string &o, unsigned major, uint64_t v) {
    const unsigned mt = major << 5;
    if (v < 24) {
        o.push_back((char)(mt | v));
    } else if (v <= 0xFF) {
        o.push_back((char)(mt | 24));
        o.push_back((char)v);
    } else if (v <= 0xFFFF) {
        o.push_back((char)(mt | 25));
        o.push_back((char)(v >> 8));
        o.push_back((char)v);
    } else if (v <= 0xFFFFFFFFull) {
        o.push_back((char)(mt | 26));
        for (int s = 24; s >= 0; s -= 8) o.push_back((char)(v >> s));
    } else {
        o.push_back((char)(mt | 27));
        for (int s = 56; s >= 0; s -= 8) o.push_back((char)(v >> s));
    }
}

void put_cbor_int(std::string &o, int64_t v) {
    if (v >= 0) put_cbor_head(o, 0, (uint64_t)v); else put_cbor_head(o, 1, (uint64_t)(-(v + 1)));
}

// cbor2 encode_float (canonical=False): NaN and the infinities as float16, everything else as float64
void put_cbor_float(std::string &o, double x) {
    if (std::isnan(x)) {
        o.append("\xf9\x7e\x00", 3);
    } else if (std::isinf(x)) {
        o.append(x > 0 ? "\xf9\x7c\x00" : "\xf9\xfc\x00", 3);
    } else {
        uint64_t bits;
        std::memcpy(&bits, &x, 8);
        o.push_back((char)0xFB);
        for (int s = 56; s >= 0; s -= 8) o.push_back((char)(bits >> s));
    }
}

void put_cbor_text(std::string &o, const char *s) {
    const size_t n = std::strlen(s);
    put_cbor_head(o, 3, n);
    o.append(s, n);
}

// cbor2 encode_datetime with datetime_as_timestamp: tag 1 around timegm(utctimetuple) (an int when the
// microsecond field is zero) or timegm(...) + microsecond / 1000000 (a float)
void put_cbor_datetime(std::string &o, int64_t ts_us) {
    int64_t sec = ts_us / 1000000, us = ts_us % 1000000;
    if (us < 0) {
        us += 1000000;
        sec -= 1;
    }
    put_cbor_head(o, 6, 1);
    if (us == 0) put_cbor_int(o, sec); else put_cbor_float(o, (double)sec + (double)us / 1000000.0);
}

// cborify (consume.py:35-39): CBORTag(1337, timedelta.total_seconds())
void put_cbor_duration(std::string &o, int64_t dur_us) {
    put_cbor_head(o, 6, 1337);
    put_cbor_float(o, total_seconds(dur_us));
}

const char *const kSignalHeader[9] = {"Device", "Time", "Frequency", "Duration", "max (dBW)", "avg (dBW)", "std (dB)",
                                      "noise (dBW)", "snr (dB)"};  // __init__.py:172-182

// Rows in blocks of kRowsPerBlock, a block per task (rt_hostpar.h): every block is formatted into a string of its own by
// `one(o, i)`, the blocks' sizes give every row its offset, and the strings are copied to their places in `out` -- block order,
// so the bytes are those of one pass over all rows.
constexpr size_t kRowsPerBlock = 2048;

template <class One>
int format_blocks(size_t n, size_t reserve_per_row, char *out, size_t cap, size_t *offsets, size_t *n_bytes, One one) {
    const size_t n_blocks = (n + kRowsPerBlock - 1) / kRowsPerBlock;
    const int threads = rt::host_threads_for(n_blocks);
    // One growing string per WORKER, its blocks one behind the other (a string per block was an mmap and its page faults per
    // block: the threads queued for the address space's lock instead of formatting); where a block lies is noted per block.
    struct Where {
        int worker;
        size_t at, size;
    };
    std::vector<std::string> arena((size_t)threads);
    std::vector<Where> where(n_blocks);
    std::vector<size_t> rel(offsets ? n : 0);  // row offsets inside their block
    std::atomic<bool> failed{false};
    rt::parallel_blocks(n_blocks, threads, [&](size_t b, int worker) {
        try {
            const size_t lo = b * kRowsPerBlock, hi = std::min(n, lo + kRowsPerBlock);
            std::string &o = arena[(size_t)worker];
            if (o.capacity() == 0) o.reserve((n / (size_t)threads + kRowsPerBlock) * reserve_per_row);
            const size_t at = o.size();
            for (size_t i = lo; i < hi; ++i) {
                if (offsets) rel[i] = o.size() - at;
                one(o, i);
            }
            where[b] = Where{worker, at, o.size() - at};
        } catch (...) {
            failed.store(true);
        }
    });
    if (failed.load()) return RT_E_NOMEM;
    std::vector<size_t> base(n_blocks + 1, 0);
    for (size_t b = 0; b < n_blocks; ++b) base[b + 1] = base[b] + where[b].size;
    const size_t total = base[n_blocks];
    if (n_bytes) *n_bytes = total;
    if (offsets) {
        rt::parallel_blocks(n_blocks, threads, [&](size_t b, int) {
            const size_t lo = b * kRowsPerBlock, hi = std::min(n, lo + kRowsPerBlock);
            for (size_t i = lo; i < hi; ++i) offsets[i] = base[b] + rel[i];
        });
        offsets[n] = total;
    }
    if (cap < total || (!out && total)) return RT_E_CAPACITY;
    rt::parallel_blocks(n_blocks, threads, [&](size_t b, int) {
        if (where[b].size) std::memcpy(out + base[b], arena[(size_t)where[b].worker].data() + where[b].at, where[b].size);
    });
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_format_float_repr(double x, char *buf) { return buf ? float_repr(x, buf) : RT_E_INVALID; }

int rt_format_signals(int32_t kind, const rt_signal_row *rows, size_t n, const char *const *device_names,
                      int32_t n_devices, char *out, size_t cap, size_t *offsets, size_t *n_bytes) {
    if ((!rows && n) || (!device_names && n) || kind < RT_FORMAT_CSV || kind > RT_FORMAT_CBOR) return RT_E_INVALID;
    for (size_t i = 0; i < n; ++i)
        if (rows[i].device < 0 || rows[i].device >= n_devices) return RT_E_INVALID;
    return format_blocks(n, 160, out, cap, offsets, n_bytes, [&](std::string &o, size_t i) {
        const rt_signal_row &r = rows[i];
        const char *dev = device_names[r.device];
        const double vals[5] = {r.max_dbw, r.avg_dbw, r.std_db, r.noise_dbw, r.snr_db};
        if (kind == RT_FORMAT_CSV) {  // [csvify(v) for v in as_list] (consume.py:195)
            put_csv_text(o, dev);
            o.push_back(';');
            put_datetime(o, r.ts_us, ' ');  // str(datetime)
            o.push_back(';');
            put_float_py(o, r.frequency);
            o.push_back(';');
            put_float_py(o, total_seconds(r.duration_us));
            for (double v : vals) {
                o.push_back(';');
                put_float_py(o, v);
            }
            o += "\r\n";
        } else if (kind == RT_FORMAT_JSON) {  // json.dumps(as_dict, default=jsonify) (consume.py:141-144)
            o.push_back('{');
            put_json_text(o, kSignalHeader[0]);
            o += ": ";
            put_json_text(o, dev);
            o += ", ";
            put_json_text(o, kSignalHeader[1]);
            o += ": \"";
            put_datetime(o, r.ts_us, 'T');  // jsonify: isoformat()
            o += "\", ";
            put_json_text(o, kSignalHeader[2]);
            o += ": ";
            put_float_json(o, r.frequency);
            o += ", ";
            put_json_text(o, kSignalHeader[3]);
            o += ": ";
            put_float_json(o, total_seconds(r.duration_us));  // jsonify: total_seconds()
            for (int k = 0; k < 5; ++k) {
                o += ", ";
                put_json_text(o, kSignalHeader[4 + k]);
                o += ": ";
                put_float_json(o, vals[k]);
            }
            o.push_back('}');
        } else {  // cbor2.dumps(as_list, ...) (consume.py:154-159)
            put_cbor_head(o, 4, 9);
            put_cbor_text(o, dev);
            put_cbor_datetime(o, r.ts_us);
            put_cbor_float(o, r.frequency);
            put_cbor_duration(o, r.duration_us);
            for (double v : vals) put_cbor_float(o, v);
        }
    });
}

// Signal rows of the records of one analysis call (consume.rows_from_analysis, natively): start time and duration from the cell
// coordinates by the reference's float64 expressions and CPython's timedelta rounding (rt_core.h: start_time, run_duration,
// timedelta_us -- what the kernels decide durations with), the frequency and the five dB figures from the caller's columns (NumPy
// evaluates the reference's float32 `10 * log10` expressions: a libm here could differ in the last place, and a CSV row prints
// every digit).
int rt_signal_rows_from_records(const rt_record *rec, size_t n, int32_t nperseg, double sample_rate, const int64_t *ts_start_us,
                                int32_t n_streams, const double *frequency, const float *max_dbw, const float *avg_dbw, const float *std_db,
                                const float *noise_dbw, const float *snr_db, rt_signal_row *out) {
    if ((!rec || !out || !ts_start_us || !frequency || !max_dbw || !avg_dbw || !std_db || !noise_dbw || !snr_db) && n) return RT_E_INVALID;
    if (nperseg < 1 || !(sample_rate > 0)) return RT_E_INVALID;
    for (size_t i = 0; i < n; ++i)
        if (rec[i].stream < 0 || rec[i].stream >= n_streams) return RT_E_INVALID;
    rt::DetectParams dp{};
    dp.nperseg = nperseg;
    dp.fs = sample_rate;
    const size_t n_blocks = (n + kRowsPerBlock - 1) / kRowsPerBlock;
    rt::parallel_blocks(n_blocks, [&](size_t b) {
        const size_t lo = b * kRowsPerBlock, hi = std::min(n, lo + kRowsPerBlock);
        for (size_t i = lo; i < hi; ++i) {
            const rt_record &r = rec[i];
            rt_signal_row &o = out[i];
            o.device = r.stream;
            o.reserved = 0;
            o.ts_us = ts_start_us[r.stream] + rt::timedelta_us(rt::start_time(dp, r.start));             // analyze.py:420-423, 436
            o.duration_us = rt::timedelta_us(rt::run_duration(dp, r.start, r.end));                       // :427, :437
            o.frequency = frequency[i];
            o.max_dbw = (double)max_dbw[i];
            o.avg_dbw = (double)avg_dbw[i];
            o.std_db = (double)std_db[i];
            o.noise_dbw = (double)noise_dbw[i];
            o.snr_db = (double)snr_db[i];
        }
    });
    return RT_OK;
}

// The records a consumer sees: the ones the shadow filter passes (analyze.py:248-251), in their order, copied to `out` (room for n) --
// `rec[rec["shadowed"] == 0]` on the host threads: blocks count, a prefix over the blocks places them, blocks copy.
int rt_records_keep_unshadowed(const rt_record *rec, size_t n, rt_record *out, size_t *n_kept) {
    if ((!rec || !out) && n) return RT_E_INVALID;
    if (!n_kept) return RT_E_INVALID;
    constexpr size_t kBlock = 16384;
    const size_t n_blocks = (n + kBlock - 1) / kBlock;
    std::vector<size_t> at(n_blocks + 1, 0);
    rt::parallel_blocks(n_blocks, [&](size_t b) {
        const size_t lo = b * kBlock, hi = std::min(n, lo + kBlock);
        size_t c = 0;
        for (size_t i = lo; i < hi; ++i) c += rec[i].shadowed == 0;
        at[b + 1] = c;
    });
    for (size_t b = 0; b < n_blocks; ++b) at[b + 1] += at[b];
    rt::parallel_blocks(n_blocks, [&](size_t b) {
        const size_t lo = b * kBlock, hi = std::min(n, lo + kBlock);
        rt_record *o = out + at[b];
        for (size_t i = lo; i < hi; ++i)
            if (rec[i].shadowed == 0) *o++ = rec[i];
    });
    *n_kept = at[n_blocks];
    return RT_OK;
}

int rt_host_set_threads(int32_t n) {
    rt::host_threads_setting().store(n < 0 ? 0 : (int)n);
    return rt::host_threads_for((size_t)1 << 30);
}

int rt_format_matched(int32_t kind, const rt_matched_row *rows, const double *avgs, const uint8_t *present, size_t n,
                      const char *const *device_names, int32_t n_devices, char *out, size_t cap, size_t *offsets,
                      size_t *n_bytes) {
    if ((!rows && n) || n_devices < 0 || (n_devices && n && (!avgs || !present || !device_names)) ||
        kind < RT_FORMAT_CSV || kind > RT_FORMAT_CBOR)
        return RT_E_INVALID;
    return format_blocks(n, 96 + 24 * (size_t)n_devices, out, cap, offsets, n_bytes, [&](std::string &o, size_t i) {
        const rt_matched_row &r = rows[i];
        const double *a = avgs + i * (size_t)n_devices;
        const uint8_t *p = present + i * (size_t)n_devices;
        if (kind == RT_FORMAT_CSV) {  // as_list = [ts, frequency, duration, *avgs] (__init__.py:262-268); None -> ''
            put_datetime(o, r.ts_us, ' ');
            o.push_back(';');
            put_float_py(o, r.frequency);
            o.push_back(';');
            put_float_py(o, total_seconds(r.duration_us));
            for (int d = 0; d < n_devices; ++d) {
                o.push_back(';');
                if (p[d]) put_float_py(o, a[d]);
            }
            o += "\r\n";
        } else if (kind == RT_FORMAT_JSON) {  // header = Time, Frequency, Duration, *devices (__init__.py:252-259)
            o += "{\"Time\": \"";
            put_datetime(o, r.ts_us, 'T');
            o += "\", \"Frequency\": ";
            put_float_json(o, r.frequency);
            o += ", \"Duration\": ";
            put_float_json(o, total_seconds(r.duration_us));
            for (int d = 0; d < n_devices; ++d) {
                o += ", ";
                put_json_text(o, device_names[d]);
                o += ": ";
                if (p[d]) put_float_json(o, a[d]); else o += "null";
            }
            o.push_back('}');
        } else {
            put_cbor_head(o, 4, 3 + (uint64_t)n_devices);
            put_cbor_datetime(o, r.ts_us);
            put_cbor_float(o, r.frequency);
            put_cbor_duration(o, r.duration_us);
            for (int d = 0; d < n_devices; ++d) {
                if (p[d]) put_cbor_float(o, a[d]); else o.push_back((char)0xF6);
            }
        }
    });
}

}  // extern "C"
