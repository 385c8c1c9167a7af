"""Timeline of ONE fit from a rocprofv3 --kernel-trace CSV: per dispatch queue, start, duration, grid; plus the chip's
idle / low-occupancy time.  Usage: python tools/trace_timeline.py trace.csv[.gz] [fit_index_from_end] [-v]"""
import csv, gzip, sys, re
path = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].lstrip("-").isdigit() else 2
verbose = "-v" in sys.argv
op = gzip.open if path.endswith(".gz") else open
rows = list(csv.DictReader(op(path, "rt")))
ev = []
for r in rows:
    name = r["Kernel_Name"]
    short = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    short = re.sub(r"\(.*", "", short)
    ev.append(dict(q=int(r["Queue_Id"]), s=int(r["Start_Timestamp"]), e=int(r["End_Timestamp"]), name=short,
                   grid=int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) * int(r["Grid_Size_Z"]), wg=int(r["Workgroup_Size_X"])))
ev.sort(key=lambda a: a["s"])
# fits are delimited by scale_transpose launches (first kernel of a fit)
starts = [i for i, a in enumerate(ev) if a["name"].startswith("scale_transpose")]
i0 = starts[-which - 1]; i1 = starts[-which]
fit = ev[i0:i1]
t0 = fit[0]["s"]
tend = max(a["e"] for a in fit)
print("fit: %d dispatches, %.3f ms wall (start to last end)" % (len(fit), (tend - t0) * 1e-6))
# union of busy intervals and of 'bulk' intervals (gemm with >= 200 workgroups)
def union(iv):
    iv = sorted(iv); tot = 0; cur_s, cur_e = None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None: tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    if cur_e is not None: tot += cur_e - cur_s
    return tot
busy = union([(a["s"], a["e"]) for a in fit])
bulk = union([(a["s"], a["e"]) for a in fit if a["name"].startswith("gemm_f64") and a["grid"] >= 200])
print("any kernel running: %.3f ms; a bulk GEMM (>=200 WGs) running: %.3f ms; sum of bulk durations %.3f ms" % (
    busy * 1e-6, bulk * 1e-6, sum(a["e"] - a["s"] for a in fit if a["name"].startswith("gemm_f64") and a["grid"] >= 200) * 1e-6))
by = {}
for a in fit:
    k = (a["name"][:60], a["q"])
    d = by.setdefault(k, [0, 0.0]); d[0] += 1; d[1] += (a["e"] - a["s"]) * 1e-6
for k, v in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print("  q%d %-62s n=%4d  %8.3f ms" % (k[1], k[0], v[0], v[1]))
if verbose:
    for a in fit:
        print("%9.1f us  +%8.1f us  q%d  grid %5d  %s" % ((a["s"] - t0) * 1e-3, (a["e"] - a["s"]) * 1e-3, a["q"], a["grid"], a["name"][:70]))
