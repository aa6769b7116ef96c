#!/bin/bash
# cost of the runtime-class integral kernel (g shells): wall times and a rocprofv3 kernel trace of tools/gpu_gshell_time.py
# usage (GPU box): bash tools/profile_gshell.sh <tag>   ->  gpurun_out/<tag>/gshell_*.txt
tag=${1:-r03}
repo=$PWD
out=$repo/gpurun_out/$tag
mkdir -p $out
export GRAFT_REPO_ROOT=$repo
cd $repo && timeout 300 python tools/gpu_gshell_time.py > $out/gshell_times.txt 2>&1
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_gs
PYTHONPATH=$repo timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_gs -- python $repo/tools/gpu_gshell_time.py > /dev/null 2> /tmp/gs.err
python $repo/tools/rocpd_summary.py $(find /tmp/prof_gs -name '*.db' | head -1) > $out/gshell_kernel_trace.txt 2>&1
tail -4 $out/gshell_times.txt; head -12 $out/gshell_kernel_trace.txt | cut -c1-200
