#!/bin/bash
# kernel trace of the C4 direct SCF (arg direct) or the C5 gradient (arg grad): top kernels by total time
repo=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_dg
rocprofv3 --kernel-trace --stats -d /tmp/prof_dg -- python $repo/tools/gpu_direct_grad_now.py $1 > /tmp/dg.log 2>&1
grep -v Warn /tmp/dg.log | tail -4
python $repo/tools/rocpd_summary.py $(find /tmp/prof_dg -name "*.db" | head -1) | head -${2:-40} | cut -c1-75,88-150
