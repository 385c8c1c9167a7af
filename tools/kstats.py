import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nfit = float(sys.argv[2])
tot = 0
for r in rows[:12]:
    print("%-62s calls/fit %6.1f  ms/fit %7.3f  avg us %8.1f" % (r["Name"][:62], int(r["Calls"]) / nfit, float(r["TotalDurationNs"]) / nfit / 1e6, float(r["AverageNs"]) / 1e3))
print("sum ms/fit %.2f" % (sum(float(r["TotalDurationNs"]) for r in rows) / nfit / 1e6))
