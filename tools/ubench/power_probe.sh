#!/bin/bash
# usage: tools/ubench/power_probe.sh  (on the GPU box)
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/power_probe power_probe.hip
( for i in $(seq 1 60); do /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket Power|sclk|mclk|fclk" | tr '\n' ' '; echo; sleep 0.25; done ) > /tmp/smi.log &
SMI=$!
/tmp/power_probe
kill $SMI 2>/dev/null || true
echo "---- rocm-smi samples"
sort /tmp/smi.log | uniq -c | sort -rn | head -20
