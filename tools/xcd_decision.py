"""Same-box A/B of the XCD-aware tile order (option xcd_order): fit time, and for the dominant GEMM instantiation the
per-launch L2 fabric-side fetches (FETCH_SIZE), the MFMA-pipe busy cycles, the active cycles and the L2 hit rate, each from
its own rocprofv3 --pmc pass.  Prints one JSON object.  usage: python tools/xcd_decision.py <scratch dir>"""
import csv, glob, json, os, re, subprocess, sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = sys.argv[1]
DOM = "gemm_f64_kernel<128, 128, false, false, true"
PASSES = {"fetch": "FETCH_SIZE", "mfma": "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE", "l2": "TCC_HIT_sum TCC_MISS_sum",
          "wait": "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"}
out = {"kernel": DOM, "what": "per launch of the dominant instantiation, averaged over the launches of 4 single-stream N=8192 fits "
                              "(tools/stage_times.py); every counter group from its own rocprofv3 --pmc pass"}
env = dict(os.environ, PYTHONPATH=R, TMPDIR="/tmp")
for order in (0, 1):
    rec = {}
    ts = subprocess.run([sys.executable, os.path.join(R, "tools", "two_streams.py"), "xcd_order=%d" % order], capture_output=True,
                        text=True, env=dict(env, NSTREAMS="1,2"), cwd="/tmp").stdout
    for s, ms in re.findall(r"streams (\d): .*?-> ([\d.]+) ms/fit", ts):
        rec["ms_per_fit_streams%s" % s] = float(ms)
    for name, counters in PASSES.items():
        d = os.path.join(O, "xcd%d_%s" % (order, name))
        subprocess.run(["rocprofv3", "--pmc"] + counters.split() + ["--kernel-trace", "--output-format", "csv", "-d", d, "--",
                        sys.executable, os.path.join(R, "tools", "stage_times.py"), "xcd_order=%d" % order],
                       capture_output=True, text=True, env=env, cwd="/tmp", timeout=600)
        acc = {}
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if DOM in row.get("Kernel_Name", ""):
                    a = acc.setdefault(row["Counter_Name"], [0.0, 0])
                    a[0] += float(row["Counter_Value"]); a[1] += 1
        for k, (v, n) in acc.items():
            rec[k + "_per_launch"] = v / max(n, 1)
            rec["launches_sampled"] = n
    if "FETCH_SIZE_per_launch" in rec:
        rec["fetch_MB_per_launch_x2_corrected"] = rec["FETCH_SIZE_per_launch"] * 2048.0 / 1e6
    if "TCC_HIT_sum_per_launch" in rec:
        rec["l2_hit_rate"] = rec["TCC_HIT_sum_per_launch"] / (rec["TCC_HIT_sum_per_launch"] + rec["TCC_MISS_sum_per_launch"])
    if "SQ_VALU_MFMA_BUSY_CYCLES_per_launch" in rec and "GRBM_GUI_ACTIVE_per_launch" in rec:
        rec["mfma_busy_over_gui_active"] = rec["SQ_VALU_MFMA_BUSY_CYCLES_per_launch"] / rec["GRBM_GUI_ACTIVE_per_launch"]
    out["xcd_order=%d" % order] = rec
print(json.dumps(out, indent=1))
