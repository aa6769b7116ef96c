#!/bin/bash
# kernel trace of tools/gpu_uks_build.py (69 eager UKS builds of one C5 molecule): kernels by total time
repo=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_uk
rocprofv3 --kernel-trace --stats -d /tmp/prof_uk -- python $repo/tools/gpu_uks_build.py > /tmp/uk.log 2>&1
grep "UKS PBE" /tmp/uk.log | tail -3
python $repo/tools/rocpd_summary.py $(find /tmp/prof_uk -name "*.db" | head -1) | head -${1:-40} | cut -c1-70,88-150
