// rt_hostpar.h -- threads for the host-side sinks (rt_format.cpp, rt_match.cpp).
//
// The GPU path hands over millions of records per second (6.4 M/s at BASELINE config 2, 22 M/s at config 4); the reference's
// consumers are per-message Python (consume.py:127-199, match.py:54-82).  The native sinks work on arrays, and since round 6 on
// several cores: blocks of rows (the formatter, the row builder) or whole matchers (one per station: the matching rule is
// sequential inside a station and independent between stations) are dealt to a small pool of std::threads created per call.
// Results are assembled in block order, so the output is byte for byte what one thread produces.
#ifndef RT_HOSTPAR_H
#define RT_HOSTPAR_H

#include <atomic>
#include <cstddef>
#include <thread>
#include <vector>

namespace rt {

inline std::atomic<int> &host_threads_setting() {
    static std::atomic<int> n{0};  // 0 = automatic
    return n;
}

// threads a call over `n_blocks` blocks of work uses: the setting (rt_host_set_threads), or -- automatic -- the machine's hardware
// threads, at most 32; never more than there are blocks
inline int host_threads_for(size_t n_blocks) {
    int t = host_threads_setting().load(std::memory_order_relaxed);
    if (t <= 0) {
        const unsigned hw = std::thread::hardware_concurrency();
        t = hw ? (int)(hw > 32u ? 32u : hw) : 1;
    }
    if ((size_t)t > n_blocks) t = (int)n_blocks;
    return t < 1 ? 1 : t;
}

// fn(block, worker) for every block in 0 .. n_blocks - 1, each exactly once, on `threads` workers (0 .. threads - 1, the caller's
// thread is worker 0; threads <= host_threads_for(n_blocks)).  fn must not throw.
template <class F>
void parallel_blocks(size_t n_blocks, int threads, F fn) {
    if (threads <= 1) {
        for (size_t b = 0; b < n_blocks; ++b) fn(b, 0);
        return;
    }
    std::atomic<size_t> next{0};
    auto work = [&](int worker) {
        for (;;) {
            const size_t b = next.fetch_add(1, std::memory_order_relaxed);
            if (b >= n_blocks) break;
            fn(b, worker);
        }
    };
    std::vector<std::thread> pool;
    pool.reserve((size_t)threads - 1);
    for (int i = 1; i < threads; ++i) pool.emplace_back(work, i);
    work(0);
    for (std::thread &th : pool) th.join();
}
template <class F>
void parallel_blocks(size_t n_blocks, F fn) {
    parallel_blocks(n_blocks, host_threads_for(n_blocks), [&](size_t b, int) { fn(b); });
}

}  // namespace rt
#endif
