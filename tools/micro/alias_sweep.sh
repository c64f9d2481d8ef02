for v in base a63 a1023 a8191; do RT_ANALYZE_LIB=$PWD/pyradiotracking_amd/librt_var_$v.so python tools/ablate_large.py 256 2048000 256 2>/dev/null | tail -1; done
