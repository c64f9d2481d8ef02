// rt_diag.h -- the one gate between the product and the laboratory.
//
// The kernels carry compile-time switches from three rounds of experiments (stage ablations, cycle stamps, timing-only
// variants that produce WRONG spectra, A/B forms of a kernel) and the host side a few environment switches for A/B runs
// and fault injection.  None of them belongs in the shipped library:
//   * a build WITHOUT -DRT_DIAG (pyradiotracking_amd/build.py: librt_analyze.so) refuses every laboratory switch on the
//     command line -- a stray -D cannot silently make a "product" with wrong spectra -- and never reads the environment
//     (RT_DIAG_ENV is a null pointer; the names do not even appear in the binary);
//   * a build WITH -DRT_DIAG (tools/variant.sh; build.py: librt_analyze_diag.so, which the fault-injection test loads)
//     accepts them all.
#ifndef RT_DIAG_H
#define RT_DIAG_H

#ifdef RT_DIAG
#include <cstdlib>
#define RT_DIAG_ENV(name) std::getenv(name)
#else
#define RT_DIAG_ENV(name) (static_cast<const char *>(nullptr))
#if defined(RT_STAMPS) || defined(RT_ABLATE) || defined(RT_DETECT_ABLATE) || defined(RT_W64_ABL) || defined(RT_WG_ABL) || defined(RT_EXP_WG_HALF) || defined(RT_EXP_DMA1024) || defined(RT_EXP_WG_NOFENCE) || defined(RT_EXP_NOBAR0) || defined(RT_EXP_NOBAR1) || \
    defined(RT_EXP_ALIAS) || defined(RT_EXP_NOWIN) || defined(RT_EXP_PRIO) || defined(RT_ONE_WAVE_MAX_R3) || defined(RT_NO_PERSIST) ||                    \
    defined(RT_WG4_MAX_R3) || defined(RT_PK_R3_MASK) || defined(RT_BELOW_MAX_R3) || defined(RT_WAVE64_4096) || defined(RT_W64_PK) ||                       \
    defined(RT_W64_PREFETCH) || defined(RT_W64_Q2AHEAD) || defined(RT_EXP6) || defined(RT_EXP_U8_PK)
#error "laboratory switch (RT_STAMPS, RT_ABLATE, RT_EXP_*, RT_W64_*, RT_*_MAX_R3, ...) in a product build: add -DRT_DIAG (tools/variant.sh does)"
#endif
#endif

#endif  // RT_DIAG_H
