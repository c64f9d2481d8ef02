#!/bin/bash
# uint8 kernels of nperseg 256 with packed butterflies at three workgroups per CU (variant u8pk) against the product, one box, interleaved
out=gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$1: value', d['value'], 'ms/step', d['ms_per_step'], 'mode', c['mode'], 'parity', (d.get('parity') or {}).get('streams_mismatched'))"; }
common="--isolated-steps 0 --parity-streams 4 --other-configs off --input u8 --lanes 3"
for rep in 1 2 3; do
for lib in product u8pk; do
  if [ $lib = product ]; then unset RT_ANALYZE_LIB; else export RT_ANALYZE_LIB=$PWD/pyradiotracking_amd/librt_var_$lib.so; fi
  python3 bench.py $common --warmup 30 --steps 200 2>>$out/err.txt | line "config2 uint8 $lib" | tee -a $out/ab.txt
  python3 bench.py $common --warmup 5 --settle 20 --steps 40 --sample-rate 300000 --streams 4096 --threshold-dbw -91 2>>$out/err.txt | line "defaults uint8 -91 $lib" | tee -a $out/ab.txt
done
done
