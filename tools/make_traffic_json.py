#!/usr/bin/env python3
"""profiles/rNN_bench_pmc_traffic.json (+ rNN_bench_kernel_stats.csv) from the summary tools/pmc_traffic.sh leaves in
gpurun_out/<name>/summary.json (run: tools/pmc_traffic.sh bt -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary).
usage: make_traffic_json.py bt r03
Refuses to write when the kernel list of the --kernel-trace pass differs from the PMC passes' (records of different builds)."""
import glob, hashlib, json, os, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, rnd = sys.argv[1], sys.argv[2]
sm = json.load(open(f"{R}/gpurun_out/{name}/summary.json"))
ours = lambda names: sorted({n.split("(")[0][:100] for n in names if "kmsr" in n})
k_trace, k_fetch, k_write = ours(s["name"] for s in sm["stats"]), ours(sm["pmc"]["FETCH_SIZE"]), ours(sm["pmc"]["WRITE_SIZE"])
if not (k_trace == k_fetch == k_write):
    sys.exit(f"kernel lists differ between the trace and the PMC passes -- not the same build / command:\n trace {k_trace}\n fetch {k_fetch}\n write {k_write}")
keys = {v: [k for k in sm["pmc"]["FETCH_SIZE"] if f"conv3x3_wino4_kernel<{v}, 0>" in k] for v in ("false", "true")}
assert keys["false"] and keys["true"], "the body conv launches (conv3x3_wino4_kernel<false> = ReLU, <true> = residual) are not in the PMC pass"
px, wbytes = 16 * 64 * 256 * 256 * 4, 36 * 64 * 64 * 4      # one feature map; the pre-transformed weights (36 points)
alg = {"false": 2 * px + wbytes, "true": 3 * px + wbytes}   # ReLU launches read x, write out; residual launches also read the skip tensor
per = {}
for v, ks in keys.items():
    f, w = sm["pmc"]["FETCH_SIZE"][ks[0]], sm["pmc"]["WRITE_SIZE"][ks[0]]
    per[v] = {"launches": f["n"], "FETCH_SIZE_kb_mean": f["mean_kb"], "WRITE_SIZE_kb_mean": w["mean_kb"],
              "hbm_read_bytes_per_launch": f["mean_kb"] * 1024 * 2, "hbm_write_bytes_per_launch": w["mean_kb"] * 1024,
              "algorithmic_bytes_per_launch": alg[v]}
    per[v]["traffic_bytes_per_launch"] = per[v]["hbm_read_bytes_per_launch"] + per[v]["hbm_write_bytes_per_launch"]
    per[v]["traffic_over_algorithmic"] = per[v]["traffic_bytes_per_launch"] / alg[v]
n = per["false"]["launches"] + per["true"]["launches"]
mean = lambda key: (per["false"][key] * per["false"]["launches"] + per["true"][key] * per["true"]["launches"]) / n
src = f"{R}/kernel-modeling-super-resolution_amd/csrc/conv3x3_wino4.hip"
out = {
    "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary  (tools/pmc_traffic.sh)",
    "kernel": "conv3x3_wino4_kernel (body conv 64->64 @256x256, batch 16; <false> = ReLU launches [read x], <true> = residual launches [read x + skip]; launch-weighted mean)",
    "kernel_source": "conv3x3_wino4.hip", "kernel_source_sha256_16": hashlib.sha256(open(src, "rb").read()).hexdigest()[:16],
    "launches": n, "by_variant": per,
    "correction": "FETCH_SIZE x 1024 x 2 (gfx950: wide coalesced reads -- 16 B/lane buffer loads and buffer LDS-DMA -- are tallied at half their bytes, MI355X_MICROARCH.md); WRITE_SIZE x 1024",
    "traffic_bytes_per_launch": mean("traffic_bytes_per_launch"), "algorithmic_bytes_per_launch": mean("algorithmic_bytes_per_launch"),
    "traffic_over_algorithmic": mean("traffic_bytes_per_launch") / mean("algorithmic_bytes_per_launch"),
    "kernel_stats": sm["stats"][:8],
    "other_kernels": {k[:80]: {"FETCH_SIZE_kb": v["mean_kb"], "WRITE_SIZE_kb": sm["pmc"]["WRITE_SIZE"].get(k, {}).get("mean_kb")}
                      for k, v in sm["pmc"]["FETCH_SIZE"].items()},
}
# ---- the degradation kernel of the same command: two shapes under one kernel name, told apart by the grid size
#      (bench step: HR [16,3,1024,1024] = 48 planes; north-star unit: HR [32,5,1024,1024] = 160 planes)
byg = sm.get("pmc_by_grid")
if byg:
    fg = {k: v for k, v in byg["FETCH_SIZE"].items() if "degrade_rowstream_kernel" in k}
    # grid = waves * 64: planes x 2 strips of 128 LR columns x (256 LR rows / rows_per_chunk) chunks; the library picks 4 rows per chunk for
    # the 48-plane batch (6144 waves) and 16 for the 160-plane unit (5120 waves)
    want = {"config2": 48 * 2 * 64 * 64, "northstar": 160 * 2 * 16 * 64}
    bygrid = {int(k.rsplit("=", 1)[1]): k for k in fg}
    if all(g in bygrid for g in want.values()):
        grids = [bygrid[want["config2"]], bygrid[want["northstar"]]]
        dsrc = f"{R}/kernel-modeling-super-resolution_amd/csrc/degrade.hip"
        shapes = {"config2_hr16x3x1024x1024": (grids[0], 16 * (4 * 3 * 1024 * 1024 + 4 * 3 * 256 * 256 + 4 * 3 * 169)),
                  "northstar_hr32x5x1024x1024": (grids[-1], 32 * (4 * 5 * 1024 * 1024 + 4 * 5 * 256 * 256 + 4 * 5 * 169))}
        deg = {"kernel_source": "degrade.hip", "kernel_source_sha256_16": hashlib.sha256(open(dsrc, "rb").read()).hexdigest()[:16], "shapes": {}}
        for key, (g, algb) in shapes.items():
            f, w = fg[g], byg["WRITE_SIZE"][g]
            tb = f["mean_kb"] * 1024 * 2 + w["mean_kb"] * 1024
            deg["shapes"][key] = {"kernel_and_grid": g, "launches": f["n"], "FETCH_SIZE_kb_mean": f["mean_kb"], "WRITE_SIZE_kb_mean": w["mean_kb"],
                                  "traffic_bytes_per_launch": tb, "algorithmic_bytes_per_launch": algb, "traffic_over_algorithmic": tb / algb}
        out["degrade"] = deg
        print("degrade traffic / algorithmic:", {k: round(v["traffic_over_algorithmic"], 3) for k, v in deg["shapes"].items()})
json.dump(out, open(f"{R}/profiles/{rnd}_bench_pmc_traffic.json", "w"), indent=1)
shutil.copy(max(glob.glob(f"{R}/gpurun_out/{name}/trace/*/*_kernel_stats.csv"), key=os.path.getmtime),   # gpurun MERGES into gpurun_out/: older boxes' directories stay
            f"{R}/profiles/{rnd}_bench_kernel_stats.csv")
print(f"traffic {out['traffic_bytes_per_launch'] / 1e6:.1f} MB per launch = {out['traffic_over_algorithmic']:.3f} x algorithmic; sha {out['kernel_source_sha256_16']}")
