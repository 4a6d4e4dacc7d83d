#!/usr/bin/env python3
"""profiles/<tag>_pmc_sq_gemm.txt from the summaries `tools/r4_profiles.sh sq` leaves in gpurun_out/r4_sq: the per-kernel counter means as collected
and two derived figures per kernel.  usage: python tools/sq_profile.py [tag]"""
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "r4_sq")
out = ["SQ counters of the GEMM kernels alone (rocprofv3 --pmc, two passes per variant; tools/r4_profiles.sh sq; means per launch over the microbenchmark's launches)",
       "f32_ring0 / f32_ring1: tools/gemm_bench.py 'affine 1/3' with option gemm_ring = 0 / 1; planes: tools/planes_bench.py (np 3 = bf16x6, np 2 = f16x3)", ""]
derived = []
for var in ("f32_ring0", "f32_ring1", "planes"):
    text = open(os.path.join(src, "summary_%s.txt" % var)).read()
    out += ["==== " + var, text.rstrip(), ""]
    cur, vals = None, {}
    for line in text.splitlines():
        m = re.match(r"\s+(SQ_\w+)\s+launches\s+\d+\s+mean\s+([0-9.]+)", line)
        if m and cur:
            vals.setdefault(cur, {})[m.group(1)] = float(m.group(2))
        elif line and not line.startswith(" "):
            cur = line.strip()
    for k, v in vals.items():
        if "SQ_WAVE_CYCLES" not in v or not v.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            continue
        wps = 3 if re.search(r"rows_gemm(_ring)?_kernel<2, 2, 2, 2", k) else 2
        derived.append("%-12s %-58s waves/SIMD %d  mfma_busy_per_simd %.2f  wait_share %.2f" %
                       (var, k, wps, v["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * v["SQ_WAVE_CYCLES"] / wps), v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"]))
out += ["==== derived (MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves, SQ_VALU_MFMA_BUSY_CYCLES cycles summed over SIMDs)",
        "mfma_busy_per_simd = MFMA_BUSY_CYCLES / (4 * WAVE_CYCLES / waves_per_SIMD): the share of a SIMD's time its matrix pipe is busy while the kernel's waves are resident",
        "wait_share = WAIT_INST_ANY / WAVE_CYCLES: the share of wave time spent waiting for an instruction's operands / counters", ""] + derived
open(os.path.join(root, "profiles", "%s_pmc_sq_gemm.txt" % tag), "w").write("\n".join(out) + "\n")
print("\n".join(derived))
