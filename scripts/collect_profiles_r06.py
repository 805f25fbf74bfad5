#!/usr/bin/env python3
"""Copies what scripts/gpu_profiles_r06.sh left under gpurun_out/p6/ into profiles/r06_* (tracked) and derives the two files
bench.py reads: profiles/r06_pmc_counts.json (vector / matrix instructions per wave and buffer of the block kernel, both forms)
and profiles/r06_pmc_traffic_<form>.json (HBM bytes per launch, gfx950 FETCH_SIZE correction)."""
import glob
import json
import os
import re
import shutil

src, dst, rnd = "gpurun_out/p6", "profiles", "r06"
for f in sorted(glob.glob(src + "/*")):
    n = os.path.basename(f)
    if os.path.isdir(f) or n.endswith(".err") or n.endswith(".log") or os.path.getsize(f) == 0:
        continue
    n = n.replace("kernel_stats_", "rocprofv3_kernel_stats_")
    shutil.copy(f, os.path.join(dst, f"{rnd}_{n}"))


def counter(text, kernel, name):
    for line in text.splitlines():
        if kernel in line and re.search(r"\s%s\s" % name, line):
            return float(line.split(name)[1].split()[0])
    return None


NB = 860      # buffers per launch of the PMC passes (bench.py's default step)
counts = {"note": f"per wave and buffer of iir_block_kernel at the headline shape (1024 objects x 512 modes, R = 4: 2048 waves x {NB} buffers per launch): "
                  f"(SQ_INSTS_VALU - SQ_INSTS_MFMA) / {2048 * NB} and SQ_INSTS_MFMA / {2048 * NB}; rocprofv3 --kernel-trace --pmc, own passes",
          "forms": {}}
for form in ("block",):
    p = f"{dst}/{rnd}_pmc_summary_{form}.txt"
    if not os.path.exists(p):
        continue
    s = open(p).read()
    valu, mfma = counter(s, "iir_block_kernel", "SQ_INSTS_VALU"), counter(s, "iir_block_kernel", "SQ_INSTS_MFMA")
    waves = counter(s, "iir_block_kernel", "SQ_WAVES")
    cfg = {"objects_per_gpu": 1024, "modes": 512, "buffers_per_step": NB, "qnorm": "sample", "scenario": "impulses"}
    if valu and mfma and waves:
        wb = waves * NB
        counts["forms"][form] = {"config": cfg, "valu_per_wave_buffer": round((valu - mfma) / wb, 1), "mfma_per_wave_buffer": round(mfma / wb, 1),
                                 "lds_per_wave_buffer": round((counter(s, "iir_block_kernel", "SQ_INSTS_LDS") or 0) / wb, 1),
                                 "coexec_cycles": counter(s, "iir_block_kernel", "SQ_VALU_MFMA_COEXEC_CYCLES"),
                                 "mfma_busy_cycles": counter(s, "iir_block_kernel", "SQ_VALU_MFMA_BUSY_CYCLES"),
                                 "source": f"profiles/{rnd}_pmc_summary_{form}.txt"}
    f, w = counter(s, "iir_block_kernel", "FETCH_SIZE"), counter(s, "iir_block_kernel", "WRITE_SIZE")
    if f and w:
        t = {"source": f"profiles/{rnd}_pmc_summary_{form}.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, bench.py --steps 3 --warmup 1 --settle 0 --clock-ramp-ms 0, start gate off: PBSO_ENGINE_OPTS=stream_sync=1)",
             "kernel": "iir_block_kernel", "config": dict(cfg, form=form), "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "fetch_correction": 2.0,
             "note": "gfx950 FETCH_SIZE counts 64 B per 128-B request (MI355X_MICROARCH.md, HBM): reads doubled; WRITE_SIZE is exact for "
                     "dword-per-lane streaming stores.",
             "traffic_bytes_per_launch": int((f * 2.0 + w) * 1024)}
        json.dump(t, open(f"{dst}/{rnd}_pmc_traffic_{form}.json", "w"), indent=1)
        print(form, "traffic bytes per launch", t["traffic_bytes_per_launch"])
if counts["forms"]:
    json.dump(counts, open(f"{dst}/{rnd}_pmc_counts.json", "w"), indent=1)
    print(json.dumps(counts["forms"], indent=1))
for f in sorted(glob.glob(f"{dst}/{rnd}_bench_*.json")):
    try:
        d = json.load(open(f))
        print(f"{os.path.basename(f):60s} rt {d['realtime_x']:8.1f}  ms/step {d['ms_per_step']:7.3f}  bank {d['roofline']['kernel_ms']:6.3f} ms  frac {d['roofline']['frac']:.3f}  max_err {d.get('max_err')}")
    except Exception as ex:
        print(os.path.basename(f), "unreadable", ex)
