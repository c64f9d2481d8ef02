#!/bin/bash
# HOST-side AddressSanitizer / UBSan build of the library (device code is not instrumented: -fno-gpu-sanitize) under the GPU tests that
# drive the host paths a CPU run cannot reach -- AUTO levels, re-runs inside rt_fetch, pool growth, lane rollback, the detrend guard.
#   build (here or on the box): tools/SANITIZE.txt's hipcc line -> pyradiotracking_amd/librt_var_asan.so ;  tools/r4/asan_host_gpu.sh <tag>
out=gpurun_out/${1:-asan}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0
export LD_PRELOAD=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
export RT_ANALYZE_LIB=$PWD/pyradiotracking_amd/librt_var_asan.so
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_runner.py -q -m gpu -k "auto or pool or lanes or prefilter or runfilter or noisy or truncated or restart or calibration or detrend or every_mode or golden_iq_case or powers_of_two or unsupported or capacity or groups or whole_stream or plateaus" > $out/asan_gpu.log 2>&1
echo "exit $?" >> $out/asan_gpu.log
echo "sanitizer reports: $(grep -c 'AddressSanitizer\|runtime error' $out/asan_gpu.log); failures that are torch refusing to start its own CUDA runtime under the preloaded ASan: $(grep -c 'E .*libcaffe2_nvrtc' $out/asan_gpu.log)"; tail -5 $out/asan_gpu.log
