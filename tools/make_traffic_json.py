"""Turn the per-kernel FETCH_SIZE summary of tools/pmc_summary.py into profiles/pmc_traffic.json, the file bench.py reads
`roofline.traffic` from.  Stamped with the source hashes of the run (bench.source_sha16): bench.py refuses a file taken
with other sources.      usage (GPU box): python tools/make_traffic_json.py <FETCH_SIZE summary txt> <out.json> [nao ngrid]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import source_sha16  # noqa: E402

KEYS = [("grid_fused", ("fused_grid_kernel",)), ("jk_tiles", ("j_stream_kernel", "jk_tiles_kernel")),
        ("grid_density", ("density_lr_kernel",)), ("grid_density_dense", ("density_kernel",)),
        ("grid_vxc", ("vxc_wsd_kernel", "vxc_wsu_kernel", "vxc_ws_kernel")), ("grid_vxc_ws2", ("vxc_ws2_kernel",))]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    nao, ngrid = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (208, 353400)
    out = {}
    for ln in open(src):
        m = re.search(r"FETCH_SIZE\s+n=(\d+)\s+avg=([0-9.e+-]+)", ln)
        if not m:
            continue
        for key, subs in KEYS:
            if any(sub in ln for sub in subs) and key not in out:
                # FETCH_SIZE is in KB and, on gfx950, reports half of a wide coalesced read (MI355X_MICROARCH.md, HBM section)
                out[key] = int(float(m.group(2)) * 1024 * 2)
                break
    doc = {"_how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python bench.py --profile-mode ... (own pass, no other trace "
                   "domains); per-dispatch average; bytes = FETCH_SIZE[KB] * 1024 * 2 (gfx950: FETCH_SIZE reports half of a wide "
                   "coalesced read; calibrated in round 1 on dqc_probe_stream_read: 1.04859e6 KB for a 2 GiB read). Source: " +
                   os.path.basename(src),
           "workload": {"nao": nao, "ngrid": ngrid, "xc": "gga"}, "hbm_read_bytes_per_launch": out}
    doc.update(source_sha16())
    json.dump(doc, open(dst, "w"), indent=1)
    print(json.dumps(doc))


if __name__ == "__main__":
    main()
