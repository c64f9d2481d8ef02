#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
hipcc -O3 --offload-arch=gfx950 -o /tmp/shift_in4 tools/micro/shift_in4.hip 2>/dev/null && /tmp/shift_in4 || exit 1
bash tools/r5/fifth.sh $1
