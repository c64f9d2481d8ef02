#!/bin/bash
# SQ counter passes of the scan kernel for one geometry (run on the GPU box through gpurun):
#   tools/sq_counters.sh <tag> <nperseg> <fs> <streams>
# two rocprofv3 --pmc passes over the torch-free workload tools/profile_traffic.py -> gpurun_out/sq_<tag>.txt
tag=$1; export RT_PROF_NPERSEG=$2 RT_PROF_FS=$3 RT_PROF_STREAMS=$4 RT_PROF_CAL=1 RT_PROF_STEPS=3
out=$PWD/gpurun_out/sq_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $out/p1 -- python3 tools/profile_traffic.py > $out/p1.json 2> $out/p1.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD \
  --kernel-trace --output-format csv -d $out/p2 -- python3 tools/profile_traffic.py > $out/p2.json 2> $out/p2.err
python3 tools/pmc_summary.py $out/p1 $out/p2 > $PWD/gpurun_out/sq_$tag.txt
python3 - $out/p1 >> $PWD/gpurun_out/sq_$tag.txt <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(list)
for path in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "rt::" in r["Kernel_Name"]:
            d[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items()):
    print(f"{k[:60]:60s} DURATION_NS  n={len(v):3d} mean={sum(v)/len(v):.1f} min={min(v):.1f} max={max(v):.1f}")
PY
rm -rf $out
grep "stft_scan" $PWD/gpurun_out/sq_$tag.txt
