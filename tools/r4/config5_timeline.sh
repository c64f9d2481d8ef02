#!/bin/bash
# kernel timeline of the config-5 share (1 024 streams x 3.2 MS, nperseg 4096), one lane: what lies between two scans.   tools/r4/config5_timeline.sh <tag>
tag=$1; out=$PWD/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
d=$out/s
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $d -- python3 bench.py --workload config5 --total-streams 1024 --lanes 1 --steps 8 --warmup 2 --settle 4 --isolated-steps 0 --no-cpu-baseline --parity-streams 0 > $out/bench.json 2> $out/bench.err || exit 1
cp $(ls $d/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
python3 tools/timeline.py $(ls $d/*/*kernel_trace.csv | head -1) > $out/timeline.txt 2>&1
python3 - $(ls $d/*/*kernel_trace.csv | head -1) > $out/last_steps.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "rt::" in r["Kernel_Name"] or "rocclr" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[-40:]:
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:12.1f} us  +{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:9.1f} us  q{r.get("Queue_Id", "?"):>3}  {r["Kernel_Name"][:70]}')
PY
rm -rf $d
python3 tools/show_bench.py $out/bench.json | cut -c1-300; cat $out/timeline.txt | head -30; cat $out/last_steps.txt
