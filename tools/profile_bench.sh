#!/bin/bash
# rocprofv3 passes of the bench command, summarised on the box (the rocpd databases are too large to bring back).
# usage (GPU box): bash tools/profile_bench.sh <tag>     -> gpurun_out/<tag>_*.txt
tag=${1:-r01}
repo=$PWD
out=$repo/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt /tmp/prof_f /tmp/prof_w
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -- python $repo/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/${tag}_bench_under_rocprof.json 2> /tmp/kt.err
python $repo/tools/rocpd_summary.py $(find /tmp/prof_kt -name '*.db' | head -1) > $out/${tag}_kernel_stats_bench_steps10.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_f -- python $repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> /tmp/f.err
python $repo/tools/pmc_summary.py $(find /tmp/prof_f -name '*.db' | head -1) FETCH_SIZE > $out/${tag}_pmc_FETCH_SIZE_bench_steps3.txt 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_w -- python $repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> /tmp/w.err
python $repo/tools/pmc_summary.py $(find /tmp/prof_w -name '*.db' | head -1) WRITE_SIZE > $out/${tag}_pmc_WRITE_SIZE_bench_steps3.txt 2>&1
head -12 $out/${tag}_kernel_stats_bench_steps10.txt; head -6 $out/${tag}_pmc_FETCH_SIZE_bench_steps3.txt; head -6 $out/${tag}_pmc_WRITE_SIZE_bench_steps3.txt
