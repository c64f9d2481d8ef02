#!/bin/bash
# rocprofv3 kernel stats of the general sizes (nperseg 128, 8192, 16384, 300): tools/r5/trace_general.sh <tag>
tag=$1
bash tools/r5/trace_cfg.sh $tag n128 --lanes 1 --sample-rate 300000 --streams 4096 --nperseg 128 &&
bash tools/r5/trace_cfg.sh $tag n8192 --lanes 1 --workload config5 --total-streams 512 --nperseg 8192 &&
bash tools/r5/trace_cfg.sh $tag n16384 --lanes 1 --workload config5 --total-streams 512 --nperseg 16384 &&
bash tools/r5/trace_cfg.sh $tag n300 --lanes 1 --sample-rate 300000 --streams 4096 --nperseg 300
