#!/bin/bash
# kernel trace of the C4 direct SCF alone: per-category kernel sums against the wall time the script prints
repo=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_do
rocprofv3 --kernel-trace --stats -d /tmp/prof_do -- python $repo/tools/gpu_direct_only.py > /tmp/do.log 2>&1
grep -v "^W2\|Warn" /tmp/do.log | tail -3
python $repo/tools/direct_scf_kernel_sums.py $(find /tmp/prof_do -name "*.db" | head -1)
