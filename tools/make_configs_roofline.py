"""profiles/<tag>_configs_roofline.json from the rocprofv3 passes of tools/profile_configs.sh (gpurun_out/<tag>/): per BASELINE
config and hot kernel -- average launch duration (kernel trace), algorithmic bytes / flops, the fraction of the
roof it sits closer to, HBM bytes by FETCH_SIZE (KB x 1024 x 2 on gfx950) and MfmaUtil.
usage: python tools/make_configs_roofline.py gpurun_out/r03b profiles/r03b"""
import json, os, re, sys
src, dst = sys.argv[1], sys.argv[2]
HBM, MFMA = 8000.0, 78.6
out = {"_how": "rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc MfmaUtil (separate passes) of tools/config_step.py <config>: "
               "steady-state Fock builds dm2scp(ao_orb2dm(C_occ)) of ONE molecule on one stream; peaks 8000 GB/s, 78.6 TF fp64 MFMA",
       "configs": {}}
HOT = ("j_stream_kernel", "jk_stream_kernel", "jk_tiles_kernel", "density_lr_kernel", "vxc_wsd_kernel", "vxc_ws_kernel", "vxc_ws2_kernel",
       "vxc_wsu_kernel", "vxc_wst_kernel", "xc_kernel")
for c in ("C2", "C3", "C3pbe", "C4", "C5"):
    if not os.path.exists(os.path.join(src, c + "_step.json")):
        continue
    step = json.loads(open(os.path.join(src, c + "_step.json")).read().strip().splitlines()[-1])
    n, G, r = step["nao"], step["ngrid"], step["nocc"]
    comp = 4 if (step["xc"] or "").startswith("gga") else 1
    kern = {}
    for ln in open(os.path.join(src, c + "_kernel_stats.txt")):
        m = re.match(r"(?:void )?dqc::(\w+)(<[^>]*>)?.*?\s(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+[\d.]+%", ln)
        if m and m.group(1) in HOT and int(m.group(3)) >= 8:
            kern[m.group(1)] = {"instantiation": m.group(1) + (m.group(2) or ""), "calls": int(m.group(3)), "avg_us": float(m.group(5))}
    for fn, key, scale in (("_pmc_FETCH_SIZE.txt", "hbm_read_bytes", 2048.0), ("_pmc_MfmaUtil.txt", "mfma_util_pct", 1.0)):
        for ln in open(os.path.join(src, c + fn)):
            m = re.match(r"(?:void )?dqc::(\w+).*?n=(\d+)\s+avg=([\d.e+-]+)", ln)
            if m and m.group(1) in kern and int(m.group(2)) >= 3:
                kern[m.group(1)][key] = float(m.group(3)) * scale
    for k, v in kern.items():
        t = v["avg_us"] * 1e-6
        if k in ("j_stream_kernel", "jk_stream_kernel", "jk_tiles_kernel"):
            by, fl = float(n) ** 4 + 3 * 8.0 * n * n, (2.0 if k == "j_stream_kernel" else 4.0) * float(n) ** 4
            v["what"] = "J + K from the stored ERI tiles (RHF)" if k != "j_stream_kernel" else "J from the stored ERI tiles"
        elif k == "density_lr_kernel":
            by, fl = 8.0 * comp * G * n + 8.0 * G * (4 if comp == 4 else 1), 4.0 * G * n * r + 2.0 * comp * G * n
            v["what"] = "density (+ gradient) on the grid from the rank-n_occ factor"
        elif k == "xc_kernel":
            by, fl = 8.0 * G * (9 if comp == 4 else 3), 0.0
            v["what"] = "functional + E_xc quadrature"
        else:
            by, fl = 8.0 * comp * G * n + 8.0 * G * (5 if comp == 4 else 2) + 8.0 * n * n, 2.0 * G * n * n + 2.0 * comp * G * n
            v["what"] = "Vxc matrix"
        v["algorithmic_bytes"], v["algorithmic_flops"] = by, fl
        v["hbm_gbs"], v["tflops"] = by / t / 1e9, fl / t / 1e12
        fh, fm = v["hbm_gbs"] / HBM, v["tflops"] / MFMA
        v["bound"], v["frac"] = ("mfma", fm) if (fm > fh and k not in ("j_stream_kernel", "jk_stream_kernel", "jk_tiles_kernel", "xc_kernel")) else ("hbm", fh)
        if "hbm_read_bytes" in v:
            v["traffic_over_algorithmic"] = v["hbm_read_bytes"] / by
    out["configs"][c] = {"shape": step, "kernels": kern}
json.dump(out, open(dst + "_configs_roofline.json", "w"), indent=1)
for c, d in out["configs"].items():
    for k, v in d["kernels"].items():
        print("%-6s %-34s %9.1f us  %-4s frac %.2f  (%.0f GB/s, %.1f TF)  traffic x%.2f  MfmaUtil %s" % (
            c, v["instantiation"][:34], v["avg_us"], v["bound"], v["frac"], v["hbm_gbs"], v["tflops"],
            v.get("traffic_over_algorithmic", float("nan")), v.get("mfma_util_pct", "-")))
