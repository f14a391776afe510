#!/bin/bash
# One replayed step's timeline (tools/rocpd_timeline.py) on the GPU box.  usage: tools/timeline.sh <tag> [batch] [shape]
TAG=${1:-r00}; B=${2:-64}; S=${3:-msvd}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/tl_$TAG -o t -- python3 $R/tools/profile_step.py fp32 8 $B $S graphs > $O/tl_$TAG.log 2>&1
D=$(find $O/tl_$TAG -name "*.db" | head -1)
python3 $R/tools/rocpd_timeline.py $D -2 $O/${TAG}_step_sequence_${S}_b$B.txt > $O/${TAG}_step_timeline_${S}_b$B.json
rm -rf $O/tl_$TAG
