import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
sys.argv = [sys.argv[0]]
import bench
for rep in range(3):
    r = bench.kfold_extra(1)
    print("kfold wall %.1f ms  rmse %.6f nlpd %.6f" % (r["wall_s"] * 1e3, r["rmse_mean"], r["nlpd_mean"]), flush=True)
