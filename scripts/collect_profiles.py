#!/usr/bin/env python3
"""Copies what scripts/gpu_profiles.sh, gpu_configs.sh, latency.py and hud_sphere.py left under
gpurun_out/final/ into profiles/ (tracked), named per round, and refreshes the PMC traffic file
that bench.py reads.  Usage: python scripts/collect_profiles.py r01"""
import glob
import json
import os
import re
import shutil
import sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
src, dst = "gpurun_out/final", "profiles"
names = {
    "bench_default.json": "bench_default.json", "bench_qnorm_closed.json": "bench_qnorm_closed.json",
    "bench_qnorm_off.json": "bench_qnorm_off.json", "bench_direct.json": "bench_direct.json",
    "bench_velocity.json": "bench_velocity.json", "bench_velocity_qnorm_closed.json": "bench_velocity_qnorm_closed.json",
    "trace_gaps.txt": "trace_gaps.txt", "bench_block_f32.json": "bench_block_f32.json",
    "pmc_summary_block.txt": "pmc_summary_block_f32.txt", "pmc_summary_block_bf16.txt": "pmc_summary_bf16.txt",
    "census_block_f32.txt": "census_1024x512_block_f32.txt",
    "bench_c2_1x512.json": "bench_c2_1x512.json", "bench_c3_64x256_listener.json": "bench_c3_64x256_listener.json",
    "bench_c5_8x4096_scraping.json": "bench_c5_8x4096_scraping.json",
    "bench_c5_8x4096_scraping_hostprof.json": "bench_c5_8x4096_scraping_hostprofiles.json",
    "census.txt": "census_1024x512.txt", "env.txt": "env.txt", "pmc_summary.txt": "pmc_summary.txt",
    "microbench3.txt": "microbench3_vgpr_banks.txt", "microbench4.txt": "microbench4_body_real_cycles.txt",
    "microbench5.txt": "microbench5_mfma_coissue.txt", "realtime_latency.txt": "realtime_latency.txt",
    "hud_sphere.txt": "hud_sphere_batch_transfer.txt", "bench_2ranks_one_gpu_gloo.json": "bench_2ranks_one_gpu_gloo.json",
}
for a, b in names.items():
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, f"{rnd}_{b}"))
stats = sorted(glob.glob(src + "/rocprof_stats/*/*_kernel_stats.csv"), key=os.path.getmtime)
if stats:
    shutil.copy(stats[-1], f"{dst}/{rnd}_rocprofv3_kernel_stats.csv")
    shutil.copy(stats[-1].replace("kernel_stats", "domain_stats"), f"{dst}/{rnd}_rocprofv3_domain_stats.csv")
s = open(f"{dst}/{rnd}_pmc_summary.txt").read()
head = json.load(open(f"{dst}/{rnd}_bench_default.json"))
kern = head["roofline"]["kernel"]
f = float(re.search(r"fetch\s+.*?%s<[^>]*>\s+FETCH_SIZE\s+(\S+)" % kern, s).group(1))
w = float(re.search(r"write\s+.*?%s<[^>]*>\s+WRITE_SIZE\s+(\S+)" % kern, s).group(1))
cfg = head["config"]
t = {
    "source": f"profiles/{rnd}_pmc_summary.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, bench.py --steps 3 --warmup 1 --settle 0)",
    "kernel": kern,
    "config": {"objects_per_gpu": cfg["objects_per_gpu"], "modes": cfg["modes"], "buffers_per_step": cfg["buffers_per_step"],
               "qnorm": "sample", "form": cfg["recurrence_form"]},
    "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "fetch_correction": 2.0,
    "note": "gfx950 FETCH_SIZE counts 64 B per 128-B request (MI355X_MICROARCH.md, HBM): reads doubled; WRITE_SIZE is exact for "
            "dword-per-lane streaming stores.",
}
t["traffic_bytes_per_launch"] = int((f * t["fetch_correction"] + w) * 1024)
json.dump(t, open(f"{dst}/{rnd}_pmc_traffic.json", "w"), indent=1)
for n in ("bench_default", "bench_qnorm_off", "bench_block_f32", "bench_velocity", "bench_velocity_qnorm_closed", "bench_direct", "bench_c2_1x512", "bench_c3_64x256_listener",
          "bench_c5_8x4096_scraping"):
    try:
        d = json.load(open(f"{dst}/{rnd}_{n}.json"))
    except Exception as ex:
        print(n, "missing", ex)
        continue
    print(f"{n:28s} rt {d['realtime_x']:7.1f}  ms/step {d['ms_per_step']:7.3f}  bank {d['roofline']['kernel_ms']:6.3f} ms  frac {d['roofline']['frac']:.3f}")
print("traffic bytes per launch", t["traffic_bytes_per_launch"])
