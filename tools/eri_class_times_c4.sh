#!/bin/bash
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_e4
rocprofv3 --kernel-trace --stats -d /tmp/prof_e4 -- python $GRAFT_REPO_ROOT/tools/gpu_eri_c4.py > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/prof_e4 -name "*.db" | head -1) | grep -E "^kernel|eri_kernel" | cut -c1-60,88-160 | head -${1:-16}
python $GRAFT_REPO_ROOT/tools/eri_kernel_sum.py $(find /tmp/prof_e4 -name "*.db" | head -1) 3
