#!/bin/bash
# variance of the default bench line with the driver's flags (--steps 20 --warmup 5), same box
out=gpurun_out/${1:-r3g}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4 5 6; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --isolated-steps 10 2>>$out/err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('rep $rep: value', d['value'], 'ms/step', d['ms_per_step'], 'scan_conc', d['roofline']['kernel_ms_concurrent'], 'iso', d['roofline']['kernel_ms'])" >> $out/var.txt || exit 1
done
cat $out/var.txt
