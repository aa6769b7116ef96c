#!/bin/bash
# per-class GPU time of the ERI fill of one 20-atom cc-pVDZ molecule (rocprofv3 kernel trace, 3 fills)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_e5
rocprofv3 --kernel-trace --stats -d /tmp/prof_e5 -- python $GRAFT_REPO_ROOT/tools/gpu_eri_c5.py > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/prof_e5 -name "*.db" | head -1) | grep -E "^kernel|eri_kernel" | cut -c1-60,88-160
python $GRAFT_REPO_ROOT/tools/eri_kernel_sum.py $(find /tmp/prof_e5 -name "*.db" | head -1) 3
