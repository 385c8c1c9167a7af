"""Fill the @@PLACEHOLDERS@@ of a DESIGN.md template from a profile set (profiles/<round>_bench.json and friends).
    python tools/fill_design.py <template> <round-tag> > DESIGN.md        (used once per round; the template is not kept)"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tpl, rt = sys.argv[1], sys.argv[2]
P = os.path.join(ROOT, "profiles")
j = json.loads([l for l in open(os.path.join(P, rt + "_bench.json")) if l.startswith("{")][0])
r, rd = j["roofline"], j["roofline_detail"]
N = 8192
v = {}
v["VALUE"] = "%.1f" % j["value"]
v["MS_STEP"] = "%.2f" % j["ms_per_step"]
v["TW_FRAC"] = "%.3f" % r["timed_window_frac_of_peak"]
v["SINGLE_MS"] = "%.2f" % j["single_stream_ms_per_fit"]
v["SINGLE_FPS"] = "%.1f" % j["single_stream_fits_per_s"]
v["FRAC"] = "%.3f" % r["frac"]
v["FRAC_TF"] = "%.1f" % r["achieved"]
v["FRAC0"] = "%.3f" % rd.get("frac_sched0", float("nan"))
v["TRAFFIC"] = "%.0f" % ((r["traffic"] or 0) / 1e6)
v["SWEEP_MS"] = "%.2f" % r["cholesky_sweep_ms"]
v["SWEEP_FRAC"] = "%.3f" % r["cholesky_sweep_frac_of_peak"]
c16 = rd["cholesky_sweep_N16384"]
v["FIT16K_MS"] = "%.1f" % c16["fit_ms"]
v["SWEEP16K_MS"] = "%.1f" % c16["ms"]
v["SWEEP16K_FRAC"] = "%.3f" % c16["frac_of_peak"]
c3 = j["cfg3_seard_N16384_d64"]
v["CFG3_MS"] = "%.1f" % c3["fit_ms"]
v["CFG3_GRAD"] = "%.2f" % c3["hadamard_reduce_ms"]
v["CFG3_ASM"] = "%.2f" % c3["assembly_fused_ms"]
a = rd["assembly_full_N16384"]
v["RBF_MS"] = "%.3f" % a["ms"]
v["RBF_HBM"] = "%.3f" % a["frac_of_hbm_peak"]
v["RBF_OF_STORES"] = "%.2f" % a.get("frac_of_stores_alone", float("nan"))
s = rd["assembly_full_N16384_SEard_d64"]
v["SEARD_MS"] = "%.3f" % s["ms"]
v["SEARD_HBM"] = "%.3f" % s["frac_of_hbm_peak"]
v["SEARD_PIPE"] = "%.2f" % s["frac_of_fp64_pipe"]
c5 = j["cfg5_ep_N4096_d32"]
v["CFG5_MS"] = "%.1f" % c5["fit_ms"]
v["CFG5_SWEEP"] = "%.2f" % c5["site_sweep_ms"]
v["CFG5_FACTOR"] = "%.2f" % c5["params_ms"]
c4 = j.get("cfg4_restarts_N8192", {})
v["CFG4_FITS"] = "%d" % c4.get("fits", 0)
v["CFG4_WALL"] = "%.2f" % c4.get("wall_s", float("nan"))
v["CFG4_FPS"] = "%.1f" % c4.get("fits_per_s", float("nan"))
p = j["predict_N8192_ns65536"]
v["PRED_MS"] = "%.0f" % p["ms"]
v["PRED_DEV"] = "%.0f" % p["device_ms"]
v["PRED_FIRST"] = "%.0f" % p["first_call_ms"]
v["PRED_ALLOC"] = "%.1f" % p["first_call_scratch_alloc_ms"]
sm = j.get("predict_N8192_small_batches", {})
v["PRED1K"] = "%.1f" % sm.get("ns1000_product_form_ms", float("nan"))
v["PRED1K_SOLVE"] = "%.1f" % sm.get("ns1000_blocked_solve_ms", float("nan"))
v["PRED8K"] = "%.1f" % sm.get("ns8192_product_form_ms", float("nan"))
v["PRED8K_SOLVE"] = "%.1f" % sm.get("ns8192_blocked_solve_ms", float("nan"))
v["KFOLD_MS"] = "%.0f" % (1e3 * j.get("kfold_K10_N8192", {}).get("wall_s", float("nan")))
v["FITC_MS"] = "%.1f" % j["fitc_n131072_nu1024"]["fit_ms"]
sf = j.get("sharded_fit", {})
v["SHARD_S"] = "%.3f" % sf.get("seconds", float("nan"))
v["SHARD_FRAC"] = "%.3f" % sf.get("frac_of_peak_per_gpu", float("nan"))
cb = j["cpu_baseline"]
v["CPU_S"] = "%.1f" % (1.0 / cb["value"])
v["CPU_SANE_S"] = "%.0f" % (1.0 / cb["value_sane_linear_algebra"])
# the two CSV rows of the dominant kernel in the single-stream rocprofv3 pass
rows = list(csv.DictReader(open(os.path.join(P, rt + "_bench_streams1_kernel_stats.csv"))))
name = lambda row: row.get("Name") or row.get("KernelName") or ""
tot_ms = calls = 0.0
fits = None
for row in rows:
    nm = name(row)
    if "gemm_f64_pair_kernel<true>" in nm or "gemm_f64_kernel<128, 128, false, false, true, true>" in nm:
        tot_ms += float(row["TotalDurationNs"]) / 1e6
        calls += float(row["Calls"])
    if "hadamard_reduce_kernel" in nm:
        fits = float(row["Calls"])
if fits:
    per_fit = tot_ms / fits
    v["CSV_LINE"] = ("%.0f calls / %.1f ms over %d fits = %.3f ms per fit, %.1f launches per fit -> %.1f TF = %.3f"
                     % (calls, tot_ms, fits, per_fit, calls / fits, 5.069e11 / per_fit / 1e9, 5.069e11 / per_fit / 1e9 / 78.6))
else:
    v["CSV_LINE"] = "(kernel-stats CSV not parsed)"
t = open(tpl).read()
for k, val in v.items():
    t = t.replace("@@" + k + "@@", val)
import re
left = sorted(set(re.findall(r"@@[A-Z0-9_]+@@", t)))
if left:
    sys.stderr.write("unfilled: %s\n" % left)
sys.stdout.write(t)
