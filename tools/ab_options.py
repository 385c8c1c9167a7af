"""A/B of option sets on ONE box, alternated inside one warmed-up process (box-to-box and cold-clock effects cancel).
    python tools/ab_options.py [N=8192] [STEPS=40] [ROUNDS=3] -- "sched=0" "sched=1" "sched=1,pair_launch=0"
For every option set: ms per fit of a single dependent chain, and fits/s with two fit streams."""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygps_amd import _lib

args = sys.argv[1:]
cut = args.index("--") if "--" in args else len(args)
kv = dict(a.split("=") for a in args[:cut])
sets = args[cut + 1:] or ["sched=0", "sched=1"]
N, STEPS, ROUNDS = int(kv.get("N", 8192)), int(kv.get("STEPS", 40)), int(kv.get("ROUNDS", 3))
d = int(kv.get("d", 16))
kind = int(kv.get("kind", 0))
lib = _lib.load()
rng = np.random.RandomState(0)
x = rng.randn(N, d); w = rng.randn(d, 1)
y = (np.sin(x @ w / np.sqrt(d)) + 0.1 * rng.randn(N, 1)).ravel()
nh = 2 if kind == 0 else d + 1


def worker(ctx, steps, out, k):
    hyp = np.concatenate([np.full(nh - 1, np.log(np.sqrt(d))), [0.0]]); m = np.full(N, y.mean()); dm = np.ones((1, N))
    alpha = np.zeros(N); nlZ = np.zeros(1); g = np.zeros(nh + 2)
    for s in range(steps):
        hyp[0] = np.log(np.sqrt(d)) + 1e-4 * (s + k)
        rc = lib.pgp_exact_fit(ctx, kind, _lib.ptr(hyp), nh, 0, 0, float(np.log(0.1)), _lib.ptr(m), _lib.ptr(dm), 1, 3,
                               _lib.ptr(alpha), _lib.ptr(nlZ), _lib.ptr(g), None)
        assert rc == 0, rc
    out[k] = nlZ[0]


ctxs = []
for k in range(2):
    h = C.c_void_p()
    assert lib.pgp_init(0, C.byref(h)) == 0
    assert lib.pgp_set_data(h, _lib.ptr(x), N, d, _lib.ptr(y)) == 0
    ctxs.append(h)


# library defaults of the options a set may name: every option that ANY set of the run names goes back to its default before a
# set is applied (round 5: sets like "sched=2,tud_tile=128" followed by "sched=2" silently kept tud_tile=128 -- two tables of that
# round had to be re-read from their first alternation)
DEFAULTS = {"sched": -1, "tud_mark": 1, "tud_tile": 64, "sched2_wide": 0, "leaf_pivot": 2, "nb_outer": 0, "pair_launch": 1, "leaf_first": 0,
            "yield": 1, "eet_overlap": 3, "eet_tile": 128, "eet_first": -1, "s_tile": 0, "lookahead": 1, "xcd_order": 0, "small_tile_below": 200,
            "gram_assembly": 1, "gram_fast": 2, "s_pan": -1, "s_pan_direct": 1, "s_pan_out": 1, "publish": 1, "trsm_lean": 1, "tail_split": 1, "tur_tile": 0}
NAMED = sorted({o.split("=")[0] for sp in sets for o in sp.split(",") if o})
for k_ in NAMED:
    assert k_ in DEFAULTS, "add the default of option %r to DEFAULTS" % k_


def apply(spec, reset=False):
    for k_ in NAMED:
        for h in ctxs:
            assert lib.pgp_set_option(h, k_.encode(), DEFAULTS[k_]) == 0, k_
    for o in spec.split(","):
        if not o:
            continue
        k_, v_ = o.split("=")
        for h in ctxs:
            assert lib.pgp_set_option(h, k_.encode(), int(v_)) == 0, o


def run(S, steps):
    out = [0] * S
    ths = [threading.Thread(target=worker, args=(ctxs[k], steps, out, k)) for k in range(S)]
    t = time.perf_counter()
    [th.start() for th in ths]
    [th.join() for th in ths]
    return time.perf_counter() - t, out[0]


run(2, 6)                                     # warm clocks and pools
res = {s: {1: [], 2: []} for s in sets}
base = sets[0]
for r in range(ROUNDS):
    for s in sets:
        apply(base); apply(s)                 # every set is applied on top of the first one
        run(1, 3)
        for S in (1, 2):
            dt, v = run(S, STEPS)
            res[s][S].append((dt / (S * STEPS) * 1e3, v))
for s in sets:
    one = [a for a, _ in res[s][1]]; two = [a for a, _ in res[s][2]]
    print("%-40s single chain %s ms (median %.3f) | two streams %s fits/s (median %.1f) | nlZ %r"
          % (s, " ".join("%.3f" % v for v in one), float(np.median(one)), " ".join("%.1f" % (1e3 / v) for v in two),
             1e3 / float(np.median(two)), res[s][1][0][1]))
