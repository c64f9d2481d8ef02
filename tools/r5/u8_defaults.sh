#!/bin/bash
# the deployment case: the reference's defaults (300 kS/s, nperseg 256, 4 096 streams) from the uint8 wire format, quantisation noise
# (-90.2 dBW per bin) under and over the threshold
out=gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$1: value', d['value'], 'ms/step', d['ms_per_step'], 'mode', c['mode'], 'records', c['records_per_step'], 'fallbacks', c['fallbacks'], 'parity', (d.get('parity') or {}).get('streams_mismatched'), 'of', (d.get('parity') or {}).get('streams_checked'))"; }
common="--warmup 5 --settle 20 --isolated-steps 0 --parity-streams 8 --other-configs off --steps 40 --sample-rate 300000 --streams 4096 --input u8"
for t in -80 -89 -91 -93; do
for lanes in 1 3; do
  python3 bench.py $common --threshold-dbw $t --lanes $lanes 2>>$out/err.txt | line "defaults uint8 threshold $t lanes $lanes" | tee -a $out/ab.txt
done
done
