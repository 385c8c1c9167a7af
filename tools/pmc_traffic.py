"""Turn rocprofv3 --pmc counter CSVs (FETCH_SIZE / WRITE_SIZE passes over bench.py) into
profiles/gemm_f64_hbm_traffic.json, the per-launch HBM traffic bench.py reports as roofline.traffic.

Units/corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE are in KiB
(x1024); on gfx950 FETCH_SIZE reads exactly half the bytes of a wide coalesced streaming read (16 B/lane loads,
which is what gemm_f64 issues) -> x2; WRITE_SIZE is taken as is."""
import csv
import glob
import hashlib
import json
import os
import sys


def gemm_source_sha16():
    """Fingerprint of the sources the dominant kernel is compiled from (bench.py refuses a traffic file of another kernel)."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pygps_amd", "csrc")
    h = hashlib.sha256()
    for f in ("gemm_f64.hip", "gemm_tile.h", "common.h"):
        h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()[:16]


def load(pattern, counter, match):
    tot, n = 0.0, 0
    for f in glob.glob(pattern, recursive=True):
        for row in csv.DictReader(open(f)):
            kn = row.get("Kernel_Name", "")
            if row.get("Counter_Name") == counter and any(m_ in kn for m_ in ((match,) if isinstance(match, str) else match)):
                tot += float(row["Counter_Value"])
                n += 1
    return tot, n


if __name__ == "__main__":
    root = sys.argv[1]
    out = sys.argv[2]
    # the dominant tile code (LDS-DMA 128-tile, with or without the yield poll) behind its two entry points: one product per launch, or
    # TU_b(p) + panel p's share of E E' in one launch (gemm_f64_pair_kernel)
    DOM = ("gemm_f64_kernel<128, 128, false, false, true", "gemm_f64_pair_kernel")
    head = sys.argv[3] if len(sys.argv) > 3 else "unknown"
    f, nf = load(root + "/fetch/**/*counter_collection.csv", "FETCH_SIZE", DOM)
    w, nw = load(root + "/write/**/*counter_collection.csv", "WRITE_SIZE", DOM)
    fa, nfa = load(root + "/fetch/**/*counter_collection.csv", "FETCH_SIZE", ("gemm_f64_kernel", "gemm_f64_pair_kernel"))
    wa, nwa = load(root + "/write/**/*counter_collection.csv", "WRITE_SIZE", ("gemm_f64_kernel", "gemm_f64_pair_kernel"))
    assert nf and nw, (nf, nw)
    fetch_b = f * 1024.0 * 2.0 / nf
    write_b = w * 1024.0 / nw
    json.dump({"kernel": " + ".join(DOM), "launches_sampled": nf, "measured_on_commit": head,
               "kernel_source_sha16": gemm_source_sha16(),
               "fetch_bytes_per_launch": fetch_b, "write_bytes_per_launch": write_b,
               "bytes_per_launch": fetch_b + write_b,
               "all_gemm_f64_instantiations": {"launches_sampled": nfa, "bytes_per_launch": fa * 2048.0 / nfa + wa * 1024.0 / nwa},
               "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `python bench.py --steps 3 --windows 1 --prof-steps 1 --streams 1 "
                       "--warmup 1 --no-cpu-baseline --no-extras`, dispatches of that instantiation averaged; FETCH_SIZE x2 "
                       "(gfx950 wide-load correction, MI355X_MICROARCH.md HBM section), KiB -> bytes"},
              open(out, "w"), indent=1)
    print(open(out).read())
