#!/bin/bash
# kernel trace + step picture of a bench workload: tools/r5/trace_cfg.sh <tag> <name> [bench flags...]
tag=$1; name=$2; shift 2
out=$PWD/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
common="--steps 10 --warmup 3 --settle 5 --isolated-steps 0 --no-cpu-baseline --parity-streams 0 --other-configs off"
d=$out/s_$name
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py $common "$@" > $out/bench_$name.json 2> $out/bench_$name.err || { echo "FAILED $name"; tail -5 $out/bench_$name.err; exit 1; }
cp $(ls $d/*/*kernel_stats.csv | head -1) $out/kernel_stats_$name.csv
python3 tools/r5/steps.py $(ls $d/*/*kernel_trace.csv | head -1) > $out/steps_$name.txt 2>&1
rm -rf $d
python3 -c "
import json,sys
d=json.loads([l for l in open('$out/bench_$name.json') if l.startswith('{')][-1]); print('$name', d['value'], d['ms_per_step'], d['config']['mode'])"
