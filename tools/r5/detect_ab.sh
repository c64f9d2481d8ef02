#!/bin/bash
# detection kernels' durations (rocprofv3 kernel trace, one lane) of library variants on configs 5 (share) and 3: tools/r5/detect_ab.sh <tag> <variant>...
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "config5 --total-streams 1024" "config3"; do
for v in "$@"; do
  export RT_ANALYZE_LIB=$PWD/pyradiotracking_amd/librt_var_$v.so
  d=/tmp/dab_$v; rm -rf $d
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $d -- python3 bench.py --workload $cfg --steps 10 --warmup 3 --settle 4 --isolated-steps 0 --no-cpu-baseline --parity-streams 0 --other-configs off > /tmp/dab.json 2>/tmp/dab.err || { echo "FAILED $v"; tail -3 /tmp/dab.err; continue; }
  python3 - "$v ${cfg%% *}" $(ls $d/*/*kernel_trace.csv | head -1) /tmp/dab.json <<'PY'
import csv, sys, statistics, json
rows = [r for r in csv.DictReader(open(sys.argv[2]))]
d=json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
out = f"{sys.argv[1]:22s} ms/step {d['ms_per_step']:8.4f}"
for pat in ("stft_scan", "detect_bucket<false>", "finalize_records"):
    v = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if pat in r["Kernel_Name"])
    if v: out += f" | {pat} n={len(v)} min {v[0]:.1f} median {statistics.median(v):.1f}"
print(out)
PY
done
done
done 2>&1 | tee gpurun_out/$tag.txt
