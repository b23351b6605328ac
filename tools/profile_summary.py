"""Summarise a rocprofv3 --kernel-trace database (…_results.db) of a bench.py run into the two files kept under profiles/:
<out>_kernel_stats.csv (Name, Calls, TotalDurationNs, AverageNs, Percentage) and <out>_summary.md.

usage: profile_summary.py results.db steps_in_trace out_prefix "title line" ["extra paragraph"]
steps_in_trace = every step bench.py ran (warm-up, timed, the single-step host probes, the roofline leg); 0 = count the
image tower's pack_image_kernel launches (one per step of C2 / C3)."""
import collections
import sqlite3
import sys

FAMILY = ("gemm_bf16_", "conv3_bf16_")   # + the 3x3 halo-patch kernel (csrc/conv3.hip): same MFMA family, other staging
TFLOP_PER_STEP = 2.897       # bench.py: algorithmic_tflop_per_step of the family (strided dgrads count executed taps)
PEAK = 2500.0


def main(db_path, steps, out, title, extra=""):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, end - start from kernels").fetchall()
    by = collections.OrderedDict()
    for name, d in rows:
        e = by.setdefault(name, [0, 0])
        e[0] += 1
        e[1] += d
    if steps <= 0:
        steps = sum(c for n, (c, t) in by.items() if "pack_image" in n)
    total = sum(v[1] for v in by.values())
    order = sorted(by.items(), key=lambda kv: -kv[1][1])
    with open(out + "_kernel_stats.csv", "w") as f:
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage\n")
        for name, (c, t) in order:
            f.write(f"\"{name}\",{c},{t},{t / c:.1f},{100.0 * t / total:.3f}\n")
    fam_c = sum(c for n, (c, t) in by.items() if any(f in n for f in FAMILY))
    fam_t = sum(t for n, (c, t) in by.items() if any(f in n for f in FAMILY))
    ms = fam_t / steps / 1e6
    tf = TFLOP_PER_STEP / (ms * 1e-3)
    with open(out + "_summary.md", "w") as f:
        f.write(f"# {title}\n\n")
        if extra:
            f.write(extra.strip() + "\n\n")
        f.write(f"* steps in the trace: {steps}; kernel time per step (sum of durations): **{total / steps / 1e6:.2f} ms**, "
                f"{len(rows) / steps:.0f} launches/step\n")
        f.write(f"* dominant kernel family `gemm_bf16_*` (tiled / grouped / phase-pipelined bodies) + `conv3_bf16_kernel`: **{ms:.2f} ms/step**, {fam_c / steps:.0f} launches/step, "
                f"average launch {fam_t / fam_c / 1e3:.1f} us -> {TFLOP_PER_STEP:.2f} TFLOP / {ms:.2f} ms = **{tf:.0f} TFLOP/s** "
                f"({100.0 * tf / PEAK:.1f} % of the 2.5 PFLOP/s dense bf16 MFMA peak)\n\n")
        f.write("| ms/step | % | calls/step | avg us | kernel |\n|---|---|---|---|---|\n")
        for name, (c, t) in order[:40]:
            f.write(f"| {t / steps / 1e6:.3f} | {100.0 * t / total:.2f} | {c / steps:.1f} | {t / c / 1e3:.1f} | `{name[:110]}` |\n")
    print(open(out + "_summary.md").read()[:1500])


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else "")
