#!/bin/bash
# per-class kernel times of the derivative-ERI launches (mode 3) of a C5 gradient: rocprofv3 kernel trace of tools/gpu_grad_breakdown.py
repo=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_gc
rocprofv3 --kernel-trace --stats -d /tmp/prof_gc -- python $repo/tools/gpu_grad_breakdown.py > /tmp/gc.log 2>&1
python $repo/tools/rocpd_summary.py $(find /tmp/prof_gc -name "*.db" | head -1) | grep -E "^kernel|, 3, 1, 1>|eri_hl" | cut -c1-60,88-150 | head -${1:-45}
