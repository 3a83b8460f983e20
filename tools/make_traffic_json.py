#!/usr/bin/env python3
"""profiles/rNN_bench_pmc_traffic.json from the summary tools/pmc_traffic.sh leaves in gpurun_out/<name>/summary.json
(run: tools/pmc_traffic.sh bt -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline).  usage: make_traffic_json.py bt r02"""
import glob, hashlib, json, os, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, rnd = sys.argv[1], sys.argv[2]
sm = json.load(open(f"{R}/gpurun_out/{name}/summary.json"))
key = [k for k in sm["pmc"]["FETCH_SIZE"] if "conv3x3_winov_kernel<false, false>" in k][0]
f, w = sm["pmc"]["FETCH_SIZE"][key], sm["pmc"]["WRITE_SIZE"][key]
rd, wr = f["mean_kb"] * 1024 * 2, w["mean_kb"] * 1024
px, wbytes = 16 * 64 * 256 * 256 * 4, 64 * 64 * 16 * 4      # one feature map; the pre-transformed weights
alg = (16 * (2 * px + wbytes) + 17 * (3 * px + wbytes)) / 33   # 16 ReLU launches (read x, write out) + 17 residual launches (+ skip read)
src = f"{R}/kernel-modeling-super-resolution_amd/csrc/conv3x3_wino_vreg.hip"
out = {
    "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline  (tools/pmc_traffic.sh)",
    "kernel": "conv3x3_winov_kernel<false, false> (body conv 64->64 @256x256, batch 16; mean over ReLU launches [read x] and residual launches [read x + skip])",
    "kernel_source": "conv3x3_wino_vreg.hip", "kernel_source_sha256_16": hashlib.sha256(open(src, "rb").read()).hexdigest()[:16],
    "launches": f["n"], "FETCH_SIZE_kb_mean": f["mean_kb"], "WRITE_SIZE_kb_mean": w["mean_kb"],
    "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
    "correction": "FETCH_SIZE x 1024 x 2 (gfx950: wide coalesced reads -- here 16 B/lane `global_load ... lds` -- are tallied at half their bytes, MI355X_MICROARCH.md); WRITE_SIZE x 1024",
    "traffic_bytes_per_launch": rd + wr, "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": (rd + wr) / alg,
    "kernel_stats": sm["stats"][:8],
    "other_kernels": {k[:80]: {"FETCH_SIZE_kb": v["mean_kb"], "WRITE_SIZE_kb": sm["pmc"]["WRITE_SIZE"].get(k, {}).get("mean_kb")}
                      for k, v in sm["pmc"]["FETCH_SIZE"].items()},
}
json.dump(out, open(f"{R}/profiles/{rnd}_bench_pmc_traffic.json", "w"), indent=1)
shutil.copy(glob.glob(f"{R}/gpurun_out/{name}/trace/*/*_kernel_stats.csv")[0], f"{R}/profiles/{rnd}_bench_kernel_stats.csv")
print(f"traffic {out['traffic_bytes_per_launch'] / 1e6:.1f} MB per launch = {out['traffic_over_algorithmic']:.3f} x algorithmic; sha {out['kernel_source_sha256_16']}")
