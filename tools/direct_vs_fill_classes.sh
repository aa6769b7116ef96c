#!/bin/bash
repo=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_df
rocprofv3 --kernel-trace --stats -d /tmp/prof_df -- python $repo/tools/gpu_fill_and_direct_c4.py > /tmp/df.log 2>&1
tail -3 /tmp/df.log
python $repo/tools/direct_vs_fill_classes.py $(find /tmp/prof_df -name "*.db" | head -1) ${1:-60}
