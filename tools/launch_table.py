"""Tabulate bench.py --launch-csv: per (operand layout, tile, shape, split) the launches per step, mean HIP-event time and
the rate against both rooflines.   python tools/launch_table.py launches.csv [steps=2]"""
import collections
import csv
import sys

COMBO = ["nt", "nn", "tn", "conv", "dgrad", "wgrad", "tn-grp", "wg-grp", "nt-grp"]
CFG = {0: "128x128", 1: "128x64", 2: "64x64", 3: "stem", 4: "256x128", 5: "128x128k32", 6: "256x128k32"}


def main(path, steps=2):
    agg = collections.OrderedDict()
    for r in csv.reader(open(path)):
        cls, combo, cfg, M, N, K, batch, split, R, stride = map(int, r[:10])
        if cls > 1:
            continue
        ms, flop = float(r[10]), float(r[11])
        nbytes = 2.0 * batch * (M * K + N * K) + (4.0 if combo in (2, 5) else 2.0) * batch * M * N
        if combo == 3:
            nbytes = 2.0 * (M * K / (R * R) * stride * stride + N * K) + 2.0 * M * N
        elif combo == 4:
            nbytes = 2.0 * (M * N + N * K) + 2.0 * M * K / (R * R) / (stride * stride)
        elif combo == 5:
            nbytes = 2.0 * (M * K + N * K / (R * R) * stride * stride) + 4.0 * M * N
        if len(r) > 12 and float(r[12]) > 0:
            nbytes = float(r[12])            # grouped launch: bytes of all its problems (M = their number)
        a = agg.setdefault((combo, cfg, M, N, K, batch, split, R, stride), [0, 0.0, flop, nbytes])
        a[0] += 1
        a[1] += ms
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for v in agg.values()) / steps
    print(f"family total {tot:.3f} ms/step, {sum(v[0] for v in agg.values()) // steps} launches/step")
    print(f"{'kind':6}{'tile':11}{'M':>7}{'N':>6}{'K':>7}{'b':>4}{'split':>6}{'R/s':>5}{'n/step':>7}{'us':>8}{'ms/step':>9}{'TF/s':>7}{'GB/s':>7}{'MB':>7}")
    for (combo, cfg, M, N, K, batch, split, R, stride), (n, ms, flop, nb) in rows:
        us = ms / n * 1e3
        print(f"{COMBO[combo]:6}{CFG.get(cfg, str(cfg)):11}{M:7d}{N:6d}{K:7d}{batch:4d}{split:6d}{R:3d}/{stride}{n / steps:7.1f}{us:8.1f}{ms / steps:9.3f}"
              f"{flop / us / 1e6:7.0f}{nb / us / 1e3:7.0f}{nb / 1e6:7.1f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2)
