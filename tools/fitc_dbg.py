import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from conftest import golden
from oracle import gp_oracle as O
import pygps_amd as pyGPs
g = golden("G12_fitc_demo_default_u")
m = pyGPs.GPR_FITC(); m.setData(g["x"], g["y"])
nlZ, dnlZ, post = m.getPosterior()
x, u = g["x"], g["u"]; hyp = g["cov_hyp"]; sn2 = np.exp(2*g["lik_hyp"][0]); snu2 = 1e-6*sn2
Kuu = O.cov_matrix(O.RBF, hyp, 0, x=u, mode="train"); Ku = O.cov_matrix(O.RBF, hyp, 0, x=u, z=x, mode="cross")
Luu = np.linalg.cholesky(Kuu + snu2*np.eye(5)).T
iKuu = np.linalg.inv(Kuu + snu2*np.eye(5))
V = np.linalg.solve(Luu.T, Ku); gs = 1 + sn2 - (V*V).sum(0)
A = np.eye(5) + (V/gs) @ V.T
Sig = np.linalg.inv(Luu.T @ A @ Luu)
print("expected L\n", Sig - iKuu); print("got\n", post.L)
print("iKuu\n", iKuu); print("Sig\n", Sig)
print("ratio got/Sig", post.L/Sig); print("got + iKuu vs Sig", (post.L + iKuu)/Sig)
