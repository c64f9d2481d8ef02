export RT_PROF_MODE=dense RT_PROF_THRESHOLD_DBW=-170 RT_PROF_CAL=1 RT_PROF_STEPS=2
out=$PWD/gpurun_out/dense_pmc; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/f -- python3 tools/profile_traffic.py > $out/f.json 2> $out/f.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/s -- python3 tools/profile_traffic.py > $out/s.json 2> $out/s.err
python3 tools/pmc_summary.py $out/f $out/s | grep "detect_dense\|stft_scan<1, 1"
