// rt_match.cpp -- host implementation of include/rt_match.h.
//
// Follows the reference's SignalMatcher.add (radiotracking/match.py:54-82) and the
// MatchingSignal arithmetic (radiotracking/__init__.py:293-406) on plain records.
// datetime / timedelta values are whole microseconds there, so int64 arithmetic
// reproduces every comparison exactly; frequency and avg stay float64.
#include <hip/hip_runtime.h>  // hipcc compiles this file as HIP too; rt_core.h needs its qualifiers

#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/rt_analyze.h"
#include "../../include/rt_match.h"
#include "rt_core.h"
#include "rt_hostpar.h"

namespace {

struct Member {
    int32_t device;
    int64_t ts_us, dur_us;
    double freq, avg;
};

// One MatchingSignal: members in insertion order (a dict keyed by device) and the three
// derived properties, refreshed whenever the member set changes.
struct Group {
    std::vector<Member> members;
    int64_t ts_us = 0;   // min(sig.ts)               __init__.py:305-313
    int64_t dur_us = 0;  // max(sig.duration)         __init__.py:294-302
    double freq = 0.0;   // statistics.median(freqs)  __init__.py:316-324

    void refresh() {
        ts_us = members[0].ts_us;
        dur_us = members[0].dur_us;
        std::vector<double> f;
        f.reserve(members.size());
        for (const Member &m : members) {
            ts_us = std::min(ts_us, m.ts_us);
            dur_us = std::max(dur_us, m.dur_us);
            f.push_back(m.freq);
        }
        // statistics.median: middle element, or the mean of the two middle ones
        std::sort(f.begin(), f.end());
        const size_t n = f.size();
        freq = (n & 1) ? f[n / 2] : (f[n / 2 - 1] + f[n / 2]) / 2.0;
    }
};

// datetime.timedelta(<unit>=x) for a float x -> whole microseconds (CPython accum(): integer part
// exactly, fraction scaled in double, leftover rounded half to even against the parity of the sum)
int64_t timedelta_units_us(double x, int64_t us_per_unit) {
    double ip;
    const double frac = std::modf(x, &ip);
    int64_t us = (int64_t)ip * us_per_unit;
    if (frac == 0.0) return us;
    double ip2;
    const double left = std::modf((double)us_per_unit * frac, &ip2);
    us += (int64_t)ip2;
    if (left != 0.0) {
        double whole = std::round(left);
        if (std::fabs(whole - left) == 0.5) {
            const int odd = (int)(us & 1LL);
            whole = 2.0 * std::round((left + odd) * 0.5) - odd;
        }
        us += (int64_t)whole;
    }
    return us;
}

// timedelta / 2: CPython divide_nearest -- quotient rounded half to even
int64_t half_us_nearest_even(int64_t us) {
    int64_t q = us / 2, r = us % 2;  // truncating
    if (r < 0) {
        q -= 1;
        r += 2;
    }
    // r in {0, 1}: remainder 1 of 2 is the exact tie -> to even
    if (r == 1 && (q & 1)) q += 1;
    return q;
}

}  // namespace

struct rt_matcher {
    rt_match_config cfg{};
    int64_t timeout_us = 0;
    int64_t time_diff_us = 0;
    int64_t half_dd_us = 0;
    bool has_dd = false;
    double half_bw = 0.0;
    std::vector<Group> groups;  // `_matched`, in list order
    std::string err;
};

namespace {

// MatchingSignal.has_member (__init__.py:337-387)
bool has_member(const rt_matcher &m, const Group &g, const rt_match_signal &s) {
    if (s.frequency - m.half_bw > g.freq) return false;                      // :366-368
    if (s.frequency + m.half_bw < g.freq) return false;                      // :369-371
    if (s.ts_us - m.time_diff_us > g.ts_us + g.dur_us) return false;         // :374-376
    if ((s.ts_us + s.duration_us) + m.time_diff_us < g.ts_us) return false;  // :378-380
    if (m.has_dd) {                                                          // :383-387
        if (s.duration_us - m.half_dd_us > g.dur_us) return false;
        if (s.duration_us + m.half_dd_us < g.dur_us) return false;
    }
    return true;
}

// MatchingSignal.add_member (__init__.py:389-406): one member per device, the louder one stays
void add_member(Group &g, const rt_match_signal &s) {
    for (Member &mb : g.members) {
        if (mb.device == s.device) {
            if (mb.avg < s.avg) {
                mb.ts_us = s.ts_us;
                mb.dur_us = s.duration_us;
                mb.freq = s.frequency;
                mb.avg = s.avg;
                g.refresh();
            }
            return;
        }
    }
    g.members.push_back(Member{s.device, s.ts_us, s.duration_us, s.frequency, s.avg});
    g.refresh();
}

void export_group(const rt_matcher &m, const Group &g, size_t k, rt_matched *out, double *out_avgs, uint8_t *out_present) {
    if (out) {
        out[k].ts_us = g.ts_us;
        out[k].duration_us = g.dur_us;
        out[k].frequency = g.freq;
        out[k].n_members = (int32_t)g.members.size();
        out[k].reserved = 0;
    }
    const int nd = m.cfg.n_devices;
    if (out_avgs)
        for (int d = 0; d < nd; ++d) out_avgs[k * nd + d] = NAN;
    if (out_present) std::memset(out_present + k * nd, 0, (size_t)nd);
    for (const Member &mb : g.members) {
        if (mb.device < 0 || mb.device >= nd) continue;  // a device outside the list has no column (:327-335)
        if (out_avgs) out_avgs[k * nd + mb.device] = mb.avg;
        if (out_present) out_present[k * nd + mb.device] = 1;
    }
}

}  // namespace

extern "C" {

int rt_match_create(const rt_match_config *cfg, rt_matcher **out) {
    if (!cfg || !out || cfg->n_devices < 0) return RT_E_INVALID;
    if (std::isnan(cfg->timeout_s) || std::isnan(cfg->time_diff_s) || std::isnan(cfg->bandwidth_hz)) return RT_E_INVALID;
    rt_matcher *m = new (std::nothrow) rt_matcher();
    if (!m) return RT_E_NOMEM;
    m->cfg = *cfg;
    m->timeout_us = rt::timedelta_us(cfg->timeout_s);      // match.py:42
    m->time_diff_us = rt::timedelta_us(cfg->time_diff_s);  // match.py:43
    m->half_bw = cfg->bandwidth_hz / 2;                    // match.py:44; `bandwidth / 2` at __init__.py:366
    // match.py:45: `timedelta(milliseconds=x) if x else None`; has_member tests `if duration_diff:` again,
    // so a value that rounds to zero microseconds disables the check as well
    if (!std::isnan(cfg->duration_diff_ms) && cfg->duration_diff_ms != 0.0) {
        const int64_t dd = timedelta_units_us(cfg->duration_diff_ms, 1000);
        m->has_dd = dd != 0;
        m->half_dd_us = half_us_nearest_even(dd);
    }
    *out = m;
    return RT_OK;
}

void rt_match_destroy(rt_matcher *m) { delete m; }

int rt_match_reset(rt_matcher *m) {
    if (!m) return RT_E_INVALID;
    m->groups.clear();
    return RT_OK;
}

int rt_match_pending_count(rt_matcher *m, size_t *n_out) {
    if (!m || !n_out) return RT_E_INVALID;
    *n_out = m->groups.size();
    return RT_OK;
}

int rt_match_add(rt_matcher *m, const rt_match_signal *sigs, size_t n, rt_matched *out, double *out_avgs,
                 uint8_t *out_present, size_t cap, size_t *n_out) {
    if (!m || (!sigs && n) || !n_out) return RT_E_INVALID;
    *n_out = 0;
    if (cap < m->groups.size() + n) {
        m->err = "rt_match_add: output capacity below pending groups + signals";
        return RT_E_CAPACITY;
    }
    size_t k = 0;
    for (size_t i = 0; i < n; ++i) {
        const rt_match_signal &s = sigs[i];
        const int64_t now = s.ts_us;  // match.py:65
        bool placed = false;
        size_t gi = 0;
        while (gi < m->groups.size()) {  // match.py:68 (a copy of the list: removing while walking is fine)
            Group &g = m->groups[gi];
            if (g.ts_us < now - m->timeout_us) {  // match.py:69-72: timed out -> consume, keep looking
                export_group(*m, g, k++, out, out_avgs, out_present);
                m->groups.erase(m->groups.begin() + (std::ptrdiff_t)gi);
                continue;
            }
            if (has_member(*m, g, s)) {  // match.py:74-77: first match wins, later groups are not visited
                add_member(g, s);
                placed = true;
                break;
            }
            ++gi;
        }
        if (!placed) {  // match.py:79-82
            Group g;
            g.members.push_back(Member{s.device, s.ts_us, s.duration_us, s.frequency, s.avg});
            g.refresh();
            m->groups.push_back(std::move(g));
        }
    }
    *n_out = k;
    return RT_OK;
}

int rt_match_pending_count_many(rt_matcher *const *ms, size_t n_matchers, size_t *n_out) {
    if ((!ms || !n_out) && n_matchers) return RT_E_INVALID;
    for (size_t k = 0; k < n_matchers; ++k) {
        if (!ms[k]) return RT_E_INVALID;
        n_out[k] = ms[k]->groups.size();
    }
    return RT_OK;
}

int rt_match_add_many(rt_matcher *const *ms, size_t n_matchers, const rt_match_signal *sigs, const size_t *sig_offsets,
                      rt_matched *out, double *out_avgs, uint8_t *out_present, const size_t *out_offsets, int32_t n_devices_max,
                      size_t *n_out) {
    if ((!ms || !sig_offsets || !out_offsets || !n_out) && n_matchers) return RT_E_INVALID;
    // everything that can be refused is checked for every matcher before any of them changes
    for (size_t k = 0; k < n_matchers; ++k) {
        rt_matcher *m = ms[k];
        if (!m || sig_offsets[k + 1] < sig_offsets[k] || out_offsets[k + 1] < out_offsets[k] || m->cfg.n_devices > n_devices_max) return RT_E_INVALID;
        const size_t n = sig_offsets[k + 1] - sig_offsets[k];
        if ((!sigs && n) || ((out_avgs || out_present) && n_devices_max < 0)) return RT_E_INVALID;
        if (out_offsets[k + 1] - out_offsets[k] < m->groups.size() + n) {
            m->err = "rt_match_add_many: output capacity below pending groups + signals";
            return RT_E_CAPACITY;
        }
    }
    // a block = kPerBlock consecutive matchers (a station's share of a call is a few signals: one task per matcher would be all overhead)
    constexpr size_t kPerBlock = 16;
    const size_t n_blocks = (n_matchers + kPerBlock - 1) / kPerBlock;
    std::vector<int> rcs(n_blocks, RT_OK);
    rt::parallel_blocks(n_blocks, [&](size_t b) {
        const size_t lo = b * kPerBlock, hi = std::min(n_matchers, lo + kPerBlock);
        for (size_t k = lo; k < hi; ++k) {
            const size_t o = out_offsets[k];
            int rc;
            try {
                rc = rt_match_add(ms[k], sigs ? sigs + sig_offsets[k] : nullptr, sig_offsets[k + 1] - sig_offsets[k], out ? out + o : nullptr,
                                  out_avgs ? out_avgs + o * (size_t)n_devices_max : nullptr, out_present ? out_present + o * (size_t)n_devices_max : nullptr,
                                  out_offsets[k + 1] - o, &n_out[k]);
            } catch (...) {
                rc = RT_E_NOMEM;
            }
            if (rc != RT_OK && rcs[b] == RT_OK) rcs[b] = rc;
        }
    });
    for (int rc : rcs)
        if (rc != RT_OK) return rc;
    return RT_OK;
}

int rt_match_pending(rt_matcher *m, rt_matched *out, double *out_avgs, uint8_t *out_present, size_t cap, size_t *n_out) {
    if (!m || !n_out) return RT_E_INVALID;
    *n_out = m->groups.size();
    if (cap < m->groups.size()) {
        m->err = "rt_match_pending: output capacity below pending groups";
        return RT_E_CAPACITY;
    }
    for (size_t k = 0; k < m->groups.size(); ++k) export_group(*m, m->groups[k], k, out, out_avgs, out_present);
    return RT_OK;
}

int rt_match_has_member(rt_matcher *m, size_t index, const rt_match_signal *sig) {
    if (!m || !sig || index >= m->groups.size()) return RT_E_INVALID;
    return has_member(*m, m->groups[index], *sig) ? 1 : 0;
}

const char *rt_match_last_error(rt_matcher *m) { return m ? m->err.c_str() : "rt_match: null handle"; }

}  // extern "C"
