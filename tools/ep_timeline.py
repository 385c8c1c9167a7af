"""Timeline of the EP block sweep from a rocprofv3 kernel trace (tools/ep_trace.sh): per sweep the resident kernel, what the bulk
stream runs beside ONE block of it, and what happens between two sweeps.  usage: python tools/ep_timeline.py <kernel_trace.csv[.gz]>"""
import csv, gzip, sys

def short(n):
    for k, s in (("ep_chain", "ep_chain_kernel (resident: 1 chain + 36 prep workgroups)"), ("ep_strip", "ep_strip_kernel"),
                 ("ep_mu_strip", "ep_mu_strip_kernel"), ("ep_wait", "ep_wait_kernel (waits for the chain's counter)"),
                 ("gemm_f64_kernel<64", "gemm_f64<64>  U = strip W"), ("gemm_f64_kernel<128", "gemm_f64<128> fold, K = 128"),
                 ("ep_site_terms", "ep_site_terms_kernel"), ("gather", "gather_strided (diag Sigma)"), ("copyBuffer", "copyBuffer"),
                 ("fillBuffer", "fillBuffer")):
        if k in n:
            return s
    return n[:48]

path = sys.argv[1]
f = gzip.open(path, "rt") if path.endswith(".gz") else open(path)
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Stream_Id"]) for r in csv.DictReader(f))
ch = [i for i, r in enumerate(rows) if "ep_chain" in r[2]]
assert len(ch) >= 4, "no EP sweeps in this trace"
last = ch[-4:]                                   # the four sweeps of the last fit
print("sweeps of the last fit (resident kernel, ms):", " ".join("%.3f" % ((rows[i][1] - rows[i][0]) / 1e6) for i in last))
i0 = last[1]; s0, e0 = rows[i0][0], rows[i0][1]
# one block in the middle of the sweep: the bulk kernels between two consecutive strip launches
strips = [r for r in rows if "ep_strip" in r[2] and s0 < r[0] < e0]
a, b = strips[15], strips[16]
print("\nbulk stream beside ONE block of the chain (block 16 of sweep 2; us from the end of strip(16)):")
for r in rows:
    if a[1] <= r[0] <= b[1] and "ep_chain" not in r[2] and r is not a:
        print("  %7.1f -> %7.1f  (%5.1f us)  %s" % ((r[0] - a[1]) / 1e3, (r[1] - a[1]) / 1e3, (r[1] - r[0]) / 1e3, short(r[2])))
print("  one block = %.1f us (strip to strip)" % ((b[1] - a[1]) / 1e3))
nxt = rows[last[2]][0]
print("\nbetween two sweeps (us from the end of the resident kernel; the next one starts at %.1f):" % ((nxt - e0) / 1e3))
for r in rows:
    if e0 - 110000 <= r[0] < nxt and "ep_chain" not in r[2]:
        print("  %7.1f -> %7.1f  (%5.1f us)  %s" % ((r[0] - e0) / 1e3, (r[1] - e0) / 1e3, (r[1] - r[0]) / 1e3, short(r[2])))
