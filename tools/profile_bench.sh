#!/bin/bash
# rocprofv3 passes of the bench workload, summarised on the box (the rocpd databases are too large to bring back).
# usage (GPU box): bash tools/profile_bench.sh <tag> [molecules]     -> gpurun_out/<tag>_*.txt, gpurun_out/<tag>_pmc_traffic.json
# --profile-mode = setup + warm-up + the timed steps only (same kernels, same shapes as the default run's timed region).
tag=${1:-r02}
nm=${2:-32}
repo=$PWD
out=$repo/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt /tmp/prof_f /tmp/prof_w /tmp/prof_m
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -- python $repo/bench.py --profile-mode --molecules $nm --steps 10 --warmup 2 --min-seconds 0 > $out/${tag}_bench_under_rocprof.json 2> /tmp/kt.err
python $repo/tools/rocpd_summary.py $(find /tmp/prof_kt -name '*.db' | head -1) > $out/${tag}_kernel_stats_bench_steps10.txt 2>&1
# the same with ONE stream: kernels do not overlap, so the profiler's average duration is the per-launch duration that bench.py
# measures with HIP events for `roofline` (with 3 streams the kernels of different molecules share the chip and stretch)
rm -rf /tmp/prof_k1
rocprofv3 --kernel-trace --stats -d /tmp/prof_k1 -- python $repo/bench.py --profile-mode --molecules $nm --steps 10 --warmup 2 --min-seconds 0 --streams 1 > $out/${tag}_bench_under_rocprof_streams1.json 2> /tmp/k1.err
python $repo/tools/rocpd_summary.py $(find /tmp/prof_k1 -name '*.db' | head -1) > $out/${tag}_kernel_stats_bench_steps10_streams1.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_f -- python $repo/bench.py --profile-mode --molecules 4 --steps 3 --warmup 1 --min-seconds 0 > /dev/null 2> /tmp/f.err
python $repo/tools/pmc_summary.py $(find /tmp/prof_f -name '*.db' | head -1) FETCH_SIZE > $out/${tag}_pmc_FETCH_SIZE_bench_steps3.txt 2>&1
# (the resident grid: points of non-zero weight -- bench.py's config.ngrid; the traffic file is only accepted for that workload)
ng=$(python -c "import json,sys; print(json.loads(open('$out/${tag}_bench_under_rocprof.json').read().strip().splitlines()[-1])['config']['ngrid'])")
python $repo/tools/make_traffic_json.py $out/${tag}_pmc_FETCH_SIZE_bench_steps3.txt $out/${tag}_pmc_traffic.json 208 $ng > /dev/null
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_w -- python $repo/bench.py --profile-mode --molecules 4 --steps 3 --warmup 1 --min-seconds 0 > /dev/null 2> /tmp/w.err
python $repo/tools/pmc_summary.py $(find /tmp/prof_w -name '*.db' | head -1) WRITE_SIZE > $out/${tag}_pmc_WRITE_SIZE_bench_steps3.txt 2>&1
rocprofv3 --kernel-trace --pmc MfmaUtil -d /tmp/prof_m -- python $repo/bench.py --profile-mode --molecules 4 --steps 3 --warmup 1 --min-seconds 0 > /dev/null 2> /tmp/m.err
python $repo/tools/pmc_summary.py $(find /tmp/prof_m -name '*.db' | head -1) MfmaUtil > $out/${tag}_pmc_MfmaUtil_bench_steps3.txt 2>&1
head -12 $out/${tag}_kernel_stats_bench_steps10.txt; head -6 $out/${tag}_pmc_FETCH_SIZE_bench_steps3.txt; cat $out/${tag}_pmc_traffic.json
# per-shape kernel times by the profiler: the shape sweep under the kernel trace
rm -rf /tmp/prof_sw
rocprofv3 --kernel-trace --stats -d /tmp/prof_sw -- python $repo/tools/shape_sweep.py $out/${tag}_shape_sweep_events.txt > /dev/null 2> /tmp/sw.err
python $repo/tools/rocpd_summary.py $(find /tmp/prof_sw -name '*.db' | head -1) | grep -E "kernel|vxc|density" | head -40 > $out/${tag}_shape_sweep_rocprof.txt 2>&1
