"""Timeline analysis of a rocprofv3 --kernel-trace database: per step, the union of kernel intervals (GPU busy), the idle
gaps between consecutive kernels and the time two streams overlap.  Usage: trace_timeline.py results.db [steps_in_trace]"""
import sqlite3
import sys


def main(path, nsteps):
    db = sqlite3.connect(path)
    rows = db.execute("select start, end, name, stream_id from kernels order by start").fetchall()
    n = len(rows)
    per = n // nsteps
    # use the last full step-sized window of launches that sits inside the steady state: take rows of the last 3 steps
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0     # steps to leave out at the end (bench.py's roofline leg: 3)
    lo = n - (3 + skip) * per
    rows = rows[lo:lo + 3 * per]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    wall = (t1 - t0) / 1e6
    busy = 0.0
    cur_s, cur_e = rows[0][0], rows[0][1]
    gaps = []
    for s, e, _, _ in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    ksum = sum(e - s for s, e, _, _ in rows)
    print(f"launches {len(rows)} over 3 steps: wall {wall:.2f} ms, union-busy {busy / 1e6:.2f} ms, kernel-sum {ksum / 1e6:.2f} ms, "
          f"idle {wall - busy / 1e6:.2f} ms in {len(gaps)} gaps")
    gaps.sort()
    if gaps:
        import statistics
        print(f"gap median {statistics.median(gaps) / 1e3:.2f} us, mean {sum(gaps) / len(gaps) / 1e3:.2f} us, "
              f">20us: {sum(1 for g in gaps if g > 20000)} totalling {sum(g for g in gaps if g > 20000) / 1e6:.2f} ms; "
              f"<=20us totalling {sum(g for g in gaps if g <= 20000) / 1e6:.2f} ms")
    print(f"per step: wall {wall / 3:.2f} busy {busy / 3e6:.2f} kernel-sum {ksum / 3e6:.2f}")
    by = {}
    for s, e, name, _ in rows:
        k = name.split("(")[0][:70]
        by.setdefault(k, [0, 0.0])
        by[k][0] += 1
        by[k][1] += e - s
    for k, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:28]:
        print(f"{t / 3e6:8.3f} ms/step {c / 3:7.1f} calls {t / c / 1e3:8.1f} us  {k}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 11)
