#!/bin/bash
# default geometry (4 096 x 300 kS/s, nperseg 256) with the noise floor at -92 .. -86 dBW, same box: library variants interleaved.
#   tools/r5/ab_dg.sh <tag> <variant>...      (variant = default | name of pyradiotracking_amd/librt_var_<name>.so); env FLOORS, MODES, LANES, STREAMS_ENV
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
common="--sample-rate 300000 --streams 4096 --steps 30 --warmup 5 --settle 20 --isolated-steps 0 --no-cpu-baseline --parity-streams 8 --other-configs off"
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: value', d['value'], 'ms/step', d['ms_per_step'], 'mode', d['config']['mode'], 'records', d['config']['records_per_step'], 'parity_bad', (d.get('parity') or {}).get('streams_mismatched'))"; }
for rep in 1 2; do
for floor in ${FLOORS:--92 -90 -88 -86}; do
  for mode in ${MODES:-auto}; do
    for lanes in ${LANES:-1 2}; do
      for v in "$@"; do
        lib=$PWD/pyradiotracking_amd/librt_var_${v%%:*}.so; [ "${v%%:*}" = default ] && lib=$PWD/pyradiotracking_amd/librt_analyze.so
        envs=""; [ "$v" != "${v#*:}" ] && envs="${v#*:}"
        env $envs RT_ANALYZE_LIB=$lib timeout -k 10 300 python3 bench.py $common --lanes $lanes --mode $mode --noise-dbw $floor 2>>$out/err.txt | line "$v floor $floor $mode lanes $lanes" >> $out/ab.txt
      done
    done
  done
done
done
sort -s -k1,7 $out/ab.txt
