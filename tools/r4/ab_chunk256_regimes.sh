#!/bin/bash
# config-2 geometry with the noise floor around the -90 dBW threshold (AUTO -> chunk-bit pre-filter) and the uint8 wire format: the handle's chunk length (25) against 32, same box
tag=$1; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$1: L', c.get('segments_per_chunk'), 'value', d['value'], 'ms/step', d['ms_per_step'], 'mode', c['mode'], 'cells', c['candidate_cells_per_step'], 'records', c['records_per_step'], 'fallbacks', c['fallbacks'], 'parity', (d.get('parity') or {}).get('streams_mismatched'))"; }
for L in 0 32; do
  for n in -92 -90 -88 -86; do
    timeout -k 10 300 python3 bench.py --segs-per-chunk $L --noise-dbw $n --threshold-dbw -90 --steps 60 --warmup 10 --no-cpu-baseline --isolated-steps 0 2>>$out/err.txt | line "floor $n" >> $out/ab.txt
  done
  timeout -k 10 300 python3 bench.py --segs-per-chunk $L --noise-dbw -80 --threshold-dbw -90 --steps 60 --warmup 10 --no-cpu-baseline --isolated-steps 0 2>>$out/err.txt | line "floor -80" >> $out/ab.txt
  timeout -k 10 300 python3 bench.py --segs-per-chunk $L --input u8 --steps 100 --warmup 20 --no-cpu-baseline --isolated-steps 0 2>>$out/err.txt | line "uint8" >> $out/ab.txt
done
cat $out/ab.txt
