#!/bin/bash
# Where does eval_gto's time go?  Two ablation builds of csrc/gto.hip linked against the product objects (python -m dqc_amd.build first):
#   nostore   -- the tile flush keeps its LDS reads and barriers but issues no global store (the condition is opaque to the compiler)
#   nocompute -- the shell loop is skipped; every tile is written as zeros through the same flush path
# usage (here): bash tools/gto_ablation.sh build      -> tools/ubench/_alt/libdqc_gto_{nostore,nocompute}.so
#       (GPU box): bash tools/gto_ablation.sh run     -> timings of tools/gpu_gto_time.py with the product library and both variants
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
    mkdir -p tools/ubench/_alt
    objs=$(ls dqc_amd/csrc/_obj/*.o | grep -v "/gto.o")
    for v in nostore nocompute; do
        if [ $v = nostore ]; then printf '#define GTO_ABL_STORE (ld == 12345)\n#define GTO_ABL_COMPUTE 1\n' > dqc_amd/csrc/_gto_abl.hip
        else printf '#define GTO_ABL_STORE 1\n#define GTO_ABL_COMPUTE 0\n' > dqc_amd/csrc/_gto_abl.hip; fi
        sed -e 's|__builtin_nontemporal_store(gto_v2d|if (GTO_ABL_STORE) __builtin_nontemporal_store(gto_v2d|' \
            -e 's|for (int is = 0; is < sh.nsh; is++) {|for (int is = 0; is < (GTO_ABL_COMPUTE ? sh.nsh : 0); is++) {|' dqc_amd/csrc/gto.hip >> dqc_amd/csrc/_gto_abl.hip
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-int-to-pointer-cast -c dqc_amd/csrc/_gto_abl.hip -o /tmp/gto_$v.o
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ubench/_alt/libdqc_gto_$v.so $objs /tmp/gto_$v.o
        rm dqc_amd/csrc/_gto_abl.hip
    done
else
    for l in "" nostore nocompute; do
        if [ -n "$l" ]; then export DQC_AMD_LIB=$PWD/tools/ubench/_alt/libdqc_gto_$l.so; fi
        python tools/gpu_gto_time.py 2>&1 | grep -E "library|deriv"
    done
fi
