#!/usr/bin/env python3
"""Condense rocprofv3 outputs under gpurun_out/ into the tracked profiles/ summaries.
  gpurun_out/<stats_dir>   : rocprofv3 --kernel-trace --stats  (python3 bench.py ...)
  gpurun_out/pmc_fetch, pmc_write : rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (separate runs)
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE reports half the bytes of a
wide coalesced read stream (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for 16-byte-per-lane stores."""
import collections
import csv
import glob
import json
import re
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
stats_dir = sys.argv[2] if len(sys.argv) > 2 else "prof4"
pmc_prefix = sys.argv[3] if len(sys.argv) > 3 else "pmc"  # gpurun_out/<pmc_prefix>_fetch, _write


def short(nm):
    nm = re.sub(r"\(anonymous namespace\)::", "", nm)
    nm = re.sub(r"tdnnf::", "", nm)
    return nm.split("(")[0].replace("void ", "")


def klass(nm):
    s = short(nm)
    # TAG 1 (last template argument) = launches of the natural-gradient statistics; the 128x32 tile is theirs alone
    if (s.startswith("rows_gemm_kernel") or s.startswith("wgrad_kernel")) and (s.endswith(", 1>") or s.startswith("rows_gemm_kernel<4, 1, 1, 1")
                                                                                or s.startswith("rows_gemm_kernel<4, 1, 1, 3")):
        return "ng_skinny_gemm_f32"
    if s.startswith(("rows_gemm_group_kernel", "ng_rowdot_kernel")):
        return "ng_skinny_gemm_f32"
    if s.startswith(("ggemm_", "ng_l_", "ng_commit", "ng_set_columns", "ng_stage", "ng_fin_", "pform_combine_kernel", "stack_taps_kernel")):
        return "ng_grouped_side_chain"
    if s.startswith("rows_gemm_kernel<2, 2, 2, 2") or s.startswith("rows_gemm_kernel<2, 2, 1, 2") or s.startswith("rows_gemm_ring_kernel<2, 2, 2, 2"):
        return "rows_gemm_f32_128x128"
    if s.startswith("rows_gemm_kernel<4, 1, 1, 5") or s.startswith("rows_gemm_ring_kernel<4, 1, 1, 5"):
        return "rows_gemm_f32_128x160"
    if s.startswith("wgrad_kernel"):
        return "wgrad_f32"
    # the pre-split plane kernels: <planes, WM, WN, TM, TN, DB, ATR> -- by arithmetic, tile width, and the rows-as-K form of the weight gradients
    m = re.match(r"planes_gemm_kernel<(\d), (\d), (\d), (\d), (\d), (\w+), (\w+)>", s)
    if m:
        arith = {"2": "f16x3", "3": "bf16x6"}[m.group(1)]
        if m.group(7) == "true":
            return "planes_gemm_%s_wgrad" % arith
        return "planes_gemm_%s_256x%d" % (arith, int(m.group(3)) * int(m.group(5)) * 32)
    if s.startswith(("planes_split_kernel", "planes_sumsq", "planes_scale_kernel", "planes_pad_kernel")):
        return "planes_split"
    if s.startswith("planes_splitk_finish"):
        return "planes_gemm_splitk_finish"
    return s


stats = (glob.glob(f"gpurun_out/{stats_dir}/*/*_kernel_stats.csv") + glob.glob(f"gpurun_out/{stats_dir}/*_kernel_stats.csv"))[0]
shutil.copy(stats, f"profiles/{tag}_bench_7q_T1500_B128_kernel_stats.csv")
rows = list(csv.DictReader(open(stats)))
by = collections.defaultdict(lambda: [0, 0])
for r in rows:
    k = klass(r["Name"])
    by[k][0] += int(r["Calls"])
    by[k][1] += int(r["TotalDurationNs"])
tot = sum(v[1] for v in by.values())
with open(f"profiles/{tag}_bench_7q_T1500_B128_kernel_classes.csv", "w", newline="") as f:
    w = csv.writer(f)  # (kernel names carry commas: quoted)
    w.writerow(["kernel_class", "calls", "total_ms", "avg_us", "percent"])
    for k, (n, ns) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        w.writerow([k, n, f"{ns / 1e6:.3f}", f"{ns / n / 1e3:.1f}", f"{100.0 * ns / tot:.2f}"])


def load(d):
    f = glob.glob(f"gpurun_out/{d}/*/*_counter_collection.csv") + glob.glob(f"gpurun_out/{d}/*_counter_collection.csv")
    return list(csv.DictReader(open(f[0]))) if f else []


traffic = collections.defaultdict(lambda: dict(launches=0, fetch_kb=0.0, write_kb=0.0))
for r in load(pmc_prefix + "_fetch"):
    t = traffic[klass(r["Kernel_Name"])]
    t["launches"] += 1
    t["fetch_kb"] += float(r["Counter_Value"])
for r in load(pmc_prefix + "_write"):
    traffic[klass(r["Kernel_Name"])]["write_kb"] += float(r["Counter_Value"])
out = {}
for k, t in traffic.items():
    if t["launches"]:
        out[k] = dict(launches=t["launches"], fetch_size_kb_per_launch=t["fetch_kb"] / t["launches"],
                      write_size_kb_per_launch=t["write_kb"] / t["launches"],
                      hbm_bytes_per_launch=(2 * t["fetch_kb"] + t["write_kb"]) * 1024 / t["launches"])
# the HBM-bound passes bench.py times as one range each (roofline_hbm): bytes per pass = the bytes of ALL the pass's kernels in the
# profiled run / the launches of its primary kernel (one per pass)
PASSES = {"bn_apply_bypass": (("bn_apply_bypass_kernel",), "bn_apply_bypass_kernel"),
          "bn_relu_bwd": (("bn_relu_bwd_reduce_kernel", "bn_relu_bwd_finalize_kernel", "bn_relu_bwd_apply_kernel", "bn_relu_bwd_apply_ng_kernel", "colsum_add_kernel"),
                          "bn_relu_bwd_finalize_kernel"),
          "denominator": (("den_forward_kernel", "den_beta_kernel", "den_gamma_kernel", "den_backward_kernel", "den_mw_kernel", "den_mw_check_kernel", "den_wide"),
                          "den_gamma_kernel|den_backward_kernel"),
          "planes_split": (("planes_split_kernel", "planes_sumsq", "planes_scale_kernel", "planes_pad_kernel"), "planes_split_kernel")}
raw = collections.defaultdict(lambda: dict(launches=0, fetch_kb=0.0, write_kb=0.0))
for r in load(pmc_prefix + "_fetch"):
    t = raw[short(r["Kernel_Name"])]
    t["launches"] += 1
    t["fetch_kb"] += float(r["Counter_Value"])
for r in load(pmc_prefix + "_write"):
    raw[short(r["Kernel_Name"])]["write_kb"] += float(r["Counter_Value"])
for name, (members, primary) in PASSES.items():
    tot_b, nprim, used = 0.0, 0, []
    for k, t in raw.items():
        if k.startswith(members):
            tot_b += (2 * t["fetch_kb"] + t["write_kb"]) * 1024
            used.append(k)
        if any(k.startswith(p) for p in primary.split("|")):
            nprim += t["launches"]
    if nprim and tot_b:
        out["hbm_pass:" + name] = dict(hbm_bytes_per_pass=tot_b / nprim, passes=nprim, kernels=sorted(used))
json.dump(out, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1, sort_keys=True)
# the bench line of the profiled run (its per-class FLOPs and algorithmic bytes per step come from the launch shapes inside the
# library), the per-stream busy / alone times and the dispatch count per step
import os
for src, dst in ((f"gpurun_out/{stats_dir}_bench_line.json", f"profiles/{tag}_bench_line_of_the_profiled_run.json"),
                 (f"gpurun_out/{stats_dir}_overlap.txt", f"profiles/{tag}_stream_overlap.txt"),
                 (f"gpurun_out/{stats_dir}_dispatches.txt", f"profiles/{tag}_dispatches_per_step.txt"),
                 (f"gpurun_out/{stats_dir}_shapes.txt", f"profiles/{tag}_kernel_shapes_per_step.txt")):
    if os.path.exists(src):
        shutil.copy(src, dst)
bl = f"gpurun_out/{stats_dir}_bench_line.json"
if os.path.exists(bl):
    line = json.load(open(bl))
    steps = line["roofline"].get("event_steps") or line["steps"]  # the steps whose launches carried events
    with open(f"profiles/{tag}_class_work_per_step.csv", "w") as f:
        f.write("kernel_class,launches_per_step,event_ms_per_step,tflops_per_step,algorithmic_gb_per_step,tflop_per_s,pmc_hbm_gb_per_step,pmc_over_algorithmic\n")
        for k in line["roofline"]["all_kernels"]:
            n = k["launches"] / steps
            pm = out.get(k["kernel"], {}).get("hbm_bytes_per_launch")
            pm_step = pm * n / 1e9 if pm else None
            alg = k["algorithmic_bytes_per_step"] / 1e9
            f.write("%s,%.1f,%.3f,%.4f,%.3f,%.2f,%s,%s\n" % (k["kernel"], n, k["ms"] / steps, k["flops_per_step"] / 1e12, alg, k["tflops"],
                                                         "%.3f" % pm_step if pm_step else "", "%.3f" % (pm_step / alg) if pm_step and alg else ""))
# kernel-class tables of the other profiled runs (supernets, small shapes): gpurun_out/<stats_dir>_<name>/
for d in sorted(glob.glob(f"gpurun_out/{stats_dir}_*")):
    if not os.path.isdir(d):
        continue
    name = os.path.basename(d)[len(stats_dir) + 1:]
    st = glob.glob(f"{d}/*/*_kernel_stats.csv") + glob.glob(f"{d}/*_kernel_stats.csv")
    if not st:
        continue
    by2 = collections.defaultdict(lambda: [0, 0])
    for r in csv.DictReader(open(st[0])):
        k = klass(r["Name"])
        by2[k][0] += int(r["Calls"])
        by2[k][1] += int(r["TotalDurationNs"])
    tot2 = sum(v[1] for v in by2.values())
    with open(f"profiles/{tag}_{name}_kernel_classes.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel_class", "calls", "total_ms", "avg_us", "percent"])
        for k, (n, ns) in sorted(by2.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, n, f"{ns / 1e6:.3f}", f"{ns / n / 1e3:.1f}", f"{100.0 * ns / tot2:.2f}"])
    for suffix, dst in (("_bench_line.json", "_bench_line.json"), ("_overlap.txt", "_stream_overlap.txt"), ("_dispatches.txt", "_dispatches_per_step.txt")):
        src = f"gpurun_out/{stats_dir}_{name}{suffix}"
        if os.path.exists(src):
            shutil.copy(src, f"profiles/{tag}_{name}{dst}")
print("wrote profiles for", tag, "classes:", len(by), "pmc kernels:", len(out))
