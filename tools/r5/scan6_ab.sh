#!/bin/bash
# scan-6 (threshold-bit scan) durations of library variants, everything in order on one stream, interleaved rounds:
#   tools/r5/scan6_ab.sh <tag> <rounds> <variant>...      prints mean / min / median per run
tag=$1; rounds=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export RT_EXP_STREAMS=1
for r in $(seq $rounds); do
for v in "$@"; do
  export RT_ANALYZE_LIB=$PWD/pyradiotracking_amd/librt_var_$v.so
  common="--sample-rate 300000 --streams 4096 --steps 30 --warmup 5 --settle 20 --isolated-steps 0 --no-cpu-baseline --parity-streams 0 --other-configs off --lanes 1 --mode runfilter --noise-dbw -88"
  d=/tmp/s6_$v
  rm -rf $d
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $d -- python3 bench.py $common > /tmp/s6.json 2>/tmp/s6.err || { echo "FAILED $v"; tail -3 /tmp/s6.err; continue; }
  python3 - $v $(ls $d/*/*kernel_trace.csv | head -1) <<'PY'
import csv, sys, statistics
rows = [r for r in csv.DictReader(open(sys.argv[2]))]
for pat in ("stft_scan<1, 6", "stft_scan<1, 7", "plan_runs", "detect_bucket<false>", "finalize_records"):
    d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if pat in r["Kernel_Name"])
    d = d[: max(1, len(d))]
    if d: print(f"{sys.argv[1]:10s} {pat:22s} n={len(d):3d} min {d[0]:8.1f} median {statistics.median(d):8.1f} mean {sum(d)/len(d):8.1f}", end=" | " if pat != "finalize_records" else "\n")
PY
done
done 2>&1 | tee gpurun_out/$tag.txt
