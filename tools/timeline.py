"""Where does a step's wall time go?  Reads a rocprofv3 --kernel-trace database of an OVERLAPPED bench.py run and, per step
(a step starts at the image tower's pack_image_kernel), prints the wall time, the time at least one kernel is running, the
time exactly one / two or more streams are busy, per-stream busy time, and the largest idle gaps with the kernels around
them.   python tools/timeline.py results.db [first_step last_step]"""
import collections
import sqlite3
import sys


def main(path, lo=4, hi=10):
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end, stream_id from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if "pack_image" in r[0]]
    print(f"{len(rows)} kernels, {len(marks)} steps")
    for si in range(lo, min(hi, len(marks) - 1)):
        ks = rows[marks[si]:marks[si + 1]]
        t0, t1 = ks[0][1], rows[marks[si + 1]][1]
        ev = []
        for name, s, e, st in ks:
            ev.append((s, 1, st))
            ev.append((e, -1, st))
        ev.sort()
        active = collections.Counter()
        last = t0
        busy = one = multi = 0
        gaps = []
        for t, d, st in ev:
            n = sum(1 for v in active.values() if v > 0)
            dt = t - last
            if n >= 1:
                busy += dt
            if n == 1:
                one += dt
            if n >= 2:
                multi += dt
            if n == 0 and dt > 0:
                gaps.append((dt, last))
            active[st] += d
            last = t
        per = collections.Counter()
        for name, s, e, st in ks:
            per[st] += e - s
        wall = (t1 - t0) / 1e6
        print(f"step {si}: wall {wall:.2f} ms, busy {busy / 1e6:.2f}, one stream {one / 1e6:.2f}, >=2 streams {multi / 1e6:.2f}, idle {wall - busy / 1e6:.2f}; "
              f"kernel time per stream " + ", ".join(f"{k}: {v / 1e6:.2f}" for k, v in sorted(per.items())) + f"; {len(ks)} launches")
        gaps.sort(reverse=True)
        tot_small = sum(g for g, _ in gaps if g < 3000) / 1e6
        print(f"    {len(gaps)} idle gaps; gaps < 3 us sum to {tot_small:.2f} ms; largest:")
        for g, at in gaps[:6]:
            before = [r for r in ks if r[2] <= at][-1:]
            after = [r for r in ks if r[1] >= at + g][:1]
            print(f"      {g / 1e3:7.1f} us after {before[0][0][:60] if before else '-'} | before {after[0][0][:60] if after else '-'}")


if __name__ == "__main__":
    a = sys.argv
    main(a[1], int(a[2]) if len(a) > 2 else 4, int(a[3]) if len(a) > 3 else 10)
