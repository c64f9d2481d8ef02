#!/bin/bash
# the uint8 path against the roofline that bounds it (vector-instruction issue): PMC passes -> gpurun_out/<tag>/pmc_valu_u8.json (+ raw csv summary)
tag=${1:-r6u8}; out=$PWD/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -f $out/pmc_valu_u8.json
for block in config2_uint8 default_geometry_uint8_noise_floor; do
  export RT_PROF_BLOCK=$block
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $out/p_$block -- python3 tools/r6/u8_valu.py > $out/$block.json 2> $out/$block.err
  python3 tools/r6/u8_valu_json.py $out/p_$block $out/$block.json $out/pmc_valu_u8.json
  python3 - $out/p_$block >> $out/kernel_durations.txt <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(list)
for path in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "rt::" in r["Kernel_Name"]:
            d[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items()):
    print(f"{k[:70]:70s} DURATION_NS  n={len(v):3d} mean={sum(v)/len(v):.1f} min={min(v):.1f}")
PY
  rm -rf $out/p_$block
done
cat $out/kernel_durations.txt | grep stft_scan
