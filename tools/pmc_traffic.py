"""HBM / fabric traffic of the dominant kernel family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; they cannot
share a pass on gfx950), as a family average AND per shape class.  Units and corrections per
/opt/skills/guides/MI355X_MICROARCH.md "HBM": both counters are kilobytes; on gfx950 FETCH_SIZE tallies the 128-byte requests
of wide coalesced reads at 64 bytes, so it is doubled.

Per class: the family's launches of one training step come in a fixed order; bench.py --gemm-log writes that order with every
launch's shape (the roofline leg's records), the PMC tables hold the family's dispatches in the same order, step after step, so
dispatch i of a step is launch record i of a step.  Algorithmic bytes per launch are the class formulas of bench.py
(operands + result moved once).

Usage: pmc_traffic.py fetch.db write.db out.json [launches.csv]"""
import collections
import csv
import hashlib
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FAMILY = ("gemm_bf16_", "conv3_bf16_")   # + the 3x3 halo-patch kernel (csrc/conv3.hip): same MFMA family, other staging


def build_id():
    """sha256 over the kernel sources and the C ABI header: what a traffic figure is valid for"""
    h = hashlib.sha256()
    src = os.path.join(ROOT, "multimodal-diagnosis-ham-spine_amd", "hamspine", "csrc")
    files = sorted(f for f in os.listdir(src) if f.endswith((".hip", ".h")))
    for f in files:
        h.update(f.encode())
        h.update(open(os.path.join(src, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "hamspine.h"), "rb").read())
    return h.hexdigest()[:16]


def family_rows(path):
    db = sqlite3.connect(path)
    try:
        rows = db.execute("select name, counter_value, dispatch_id from pmc_events order by dispatch_id").fetchall()
    except sqlite3.OperationalError:
        rows = [(n, v, i) for i, (n, v) in enumerate(db.execute("select name, counter_value from pmc_events").fetchall())]
    return [(n, v) for n, v, _ in rows if any(f in n for f in FAMILY)], sum(v for _, v, _ in rows), len(rows)


def classes_of(csv_path):
    import bench
    recs = []
    with open(csv_path) as f:
        for r in csv.reader(f):
            cls, combo, cfg, M, N, K, batch, split, R, stride = map(int, r[:10])
            if cls > 1:
                continue
            flop = float(r[11])
            nbytes = 2.0 * batch * (M * K + N * K) + (4.0 if combo in (2, 5) else 2.0) * batch * M * N
            if combo == 3:
                nbytes = 2.0 * (M * K / (R * R) * stride * stride + N * K) + 2.0 * M * N
            elif combo == 4:
                nbytes = 2.0 * (M * N + N * K) + 2.0 * M * K / (R * R) / (stride * stride)
            elif combo == 5:
                nbytes = 2.0 * (M * K + N * K / (R * R) * stride * stride) + 4.0 * M * N
            name = bench.shape_class(combo, M, N, K, R)
            if combo >= 6:
                nbytes = float(r[12])
                name = {6: "pointwise_wgrad_grouped", 7: "conv_wgrad_grouped", 8: "bert_wgrad_grouped"}[combo]
            recs.append((name, nbytes, flop))
    return recs


def main(fetch_db, write_db, out, launches=None):
    f_rows, f_all, n_all = family_rows(fetch_db)
    w_rows, w_all, _ = family_rows(write_db)
    assert len(f_rows) == len(w_rows), (len(f_rows), len(w_rows))
    n = len(f_rows)
    fetch_b = 2.0 * sum(v for _, v in f_rows) * 1024.0
    write_b = sum(v for _, v in w_rows) * 1024.0
    res = {
        "kernel_family": "gemm_bf16_* (tiled / grouped / phase-pipelined bodies) + conv3_bf16_kernel", "launches": n,
        "fetch_bytes_per_launch": fetch_b / n, "write_bytes_per_launch": write_b / n,
        "traffic_bytes_per_launch": (fetch_b + write_b) / n,
        "all_kernels_bytes_total": 2.0 * f_all * 1024.0 + w_all * 1024.0, "all_kernels_launches": n_all,
        "build_id": build_id(),
        "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of `HAMSPINE_OVERLAP=0 HAMSPINE_TOWER_OVERLAP=0 bench.py --steps 3 --warmup 2`; "
                  "KB -> bytes, FETCH_SIZE doubled (gfx950 128-B requests tallied at 64 B)",
    }
    if launches:
        recs = classes_of(launches)
        per_step = len(recs) // 2                       # the roofline leg records two steps
        recs = recs[:per_step]
        if per_step and n % per_step == 0:
            steps = n // per_step
            agg = collections.OrderedDict()
            for i in range(n):
                name, alg, flop = recs[i % per_step]
                a = agg.setdefault(name, {"launches": 0, "fetch": 0.0, "write": 0.0, "alg": 0.0})
                a["launches"] += 1
                a["fetch"] += 2.0 * f_rows[i][1] * 1024.0
                a["write"] += w_rows[i][1] * 1024.0
                a["alg"] += alg
            res["by_class"] = {
                k: {"launches_per_step": v["launches"] // steps, "fabric_mb_per_step": round((v["fetch"] + v["write"]) / steps / 1e6, 1),
                    "fetch_mb_per_step": round(v["fetch"] / steps / 1e6, 1), "write_mb_per_step": round(v["write"] / steps / 1e6, 1),
                    "algorithmic_mb_per_step": round(v["alg"] / steps / 1e6, 1),
                    "ratio": round((v["fetch"] + v["write"]) / max(v["alg"], 1.0), 2)}
                for k, v in agg.items()}
            tot_alg = sum(v["alg"] for v in agg.values()) / steps
            res["family_fabric_gb_per_step"] = round((fetch_b + write_b) / steps / 1e9, 2)
            res["family_algorithmic_gb_per_step"] = round(tot_alg / 1e9, 2)
            res["family_ratio"] = round((fetch_b + write_b) / steps / max(tot_alg, 1.0), 2)
        else:
            res["by_class_error"] = f"{n} family dispatches are not a multiple of the {per_step} launches of a step"
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:5])
