"""Summarise a per-launch GEMM log (bench.py --gemm-log): per shape launches, time, TFLOP/s and the fraction of the
HBM roofline its algorithmic bytes would allow.  Columns of the log: cls,combo,cfg,M,N,K,batch,split,R,stride,ms."""
import collections
import sys

COMBO = {0: "A.kc B.kc (fwd)", 1: "A.kc B.rc (dgrad)", 2: "A.rc B.rc (wgrad)", 3: "conv fwd", 4: "conv dgrad", 5: "conv wgrad"}
CFG = {0: "128x128", 1: "128x64", 2: "64x64", 3: "stem", 4: "128x128x32", 5: "128x64x32"}


def main(path):
    rows = collections.OrderedDict()
    for line in open(path):
        cls, combo, cfg, M, N, K, batch, split, R, stride, ms = line.strip().split(",")
        key = (int(combo), int(cfg), int(M), int(N), int(K), int(batch), int(split), int(R), int(stride))
        r = rows.setdefault(key, [0, 0.0])
        r[0] += 1
        r[1] += float(ms)
    tot = sum(v[1] for v in rows.values())
    print(f"{len(rows)} distinct shapes, {sum(v[0] for v in rows.values())} launches, {tot:.3f} ms")
    print(f"{'ms':>7} {'n':>3} {'us/launch':>9} {'TF/s':>7} {'GB/s(alg)':>9}  shape")
    for key, (n, ms) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        combo, cfg, M, N, K, batch, split, R, stride = key
        flops = 2.0 * M * N * K * batch
        if combo in (3, 4):      # implicit GEMM: the image is read once, not R*S times
            abytes = M * (K // max(R * R, 1)) * 2 * (stride * stride if combo == 3 else 1)
        elif combo == 5:
            abytes = M * K * 2
        else:
            abytes = M * K * 2
        bbytes = N * K * 2 if combo != 5 else (N // max(R * R, 1)) * K * 2
        dbytes = M * N * 2
        byt = (abytes + bbytes + dbytes) * batch
        us = ms / n * 1e3
        print(f"{ms:7.3f} {n:3d} {us:9.1f} {flops / (us * 1e-6) / 1e12:7.1f} {byt / (us * 1e-6) / 1e9:9.0f}  "
              f"{COMBO[combo]:18s} {CFG.get(cfg, cfg):8s} M={M} N={N} K={K} b={batch} split={split} R={R} s={stride}")


if __name__ == "__main__":
    main(sys.argv[1])
