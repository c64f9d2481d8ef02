#!/bin/bash
# detection kernels' durations + step of library variants on several workloads: tools/r5/detect_ab2.sh <tag> <variant>...
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "--workload config4 --total-streams 8192 --lanes 2" "--sample-rate 300000 --streams 4096 --lanes 1 --mode runfilter --noise-dbw -88" "--sample-rate 300000 --streams 4096 --lanes 2" "--workload config5 --total-streams 1024"; do
for v in "$@"; do
  export RT_ANALYZE_LIB=$PWD/pyradiotracking_amd/librt_var_$v.so
  d=/tmp/dab_$v; rm -rf $d
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $d -- python3 bench.py $cfg --steps 10 --warmup 3 --settle 10 --isolated-steps 0 --no-cpu-baseline --parity-streams 0 --other-configs off > /tmp/dab.json 2>/tmp/dab.err || { echo "FAILED $v"; tail -3 /tmp/dab.err; continue; }
  python3 - "$v" $(ls $d/*/*kernel_trace.csv | head -1) /tmp/dab.json <<'PY'
import csv, sys, statistics, json
rows = [r for r in csv.DictReader(open(sys.argv[2]))]
d=json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
out = f"{sys.argv[1]:8s} {d['config']['workload'][:28]:28s} {d['config']['mode']:9s} ms/step {d['ms_per_step']:8.4f}"
for pat in ("detect_bucket<false, 256", "detect_bucket<false, 1024", "detect_bucket<false>", "finalize_records"):
    v = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if pat in r["Kernel_Name"])
    if v: out += f" | {pat[14:]} n={len(v)} med {statistics.median(v):.1f}"
print(out)
PY
done
done
done 2>&1 | tee gpurun_out/$tag.txt
