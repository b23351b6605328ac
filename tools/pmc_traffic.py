"""HBM traffic of the dominant kernel family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; they cannot share
a pass on gfx950).  Units and corrections per /opt/skills/guides/MI355X_MICROARCH.md "HBM": both counters are
kilobytes; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes, so it is doubled.
Usage: pmc_traffic.py fetch.db write.db out.json"""
import json
import sqlite3
import sys

FAMILY = "gemm_bf16_"        # gemm_bf16_kernel<...>, gemm_bf16_kernel_w3<...>, gemm_bf16_grouped_kernel<...>


def per_family(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, counter_value from pmc_events").fetchall()
    fam = [v for n, v in rows if FAMILY in n]
    return sum(fam), len(fam), sum(v for _, v in rows), len(rows)


def main(fetch_db, write_db, out):
    f_kb, f_n, f_all, n_all = per_family(fetch_db)
    w_kb, w_n, w_all, _ = per_family(write_db)
    assert f_n == w_n, (f_n, w_n)
    fetch_b = 2.0 * f_kb * 1024.0
    write_b = w_kb * 1024.0
    res = {
        "kernel_family": "gemm_bf16_kernel / _kernel_w3 / _grouped_kernel", "launches": f_n,
        "fetch_bytes_per_launch": fetch_b / f_n, "write_bytes_per_launch": write_b / f_n,
        "traffic_bytes_per_launch": (fetch_b + write_b) / f_n,
        "all_kernels_bytes_total": 2.0 * f_all * 1024.0 + w_all * 1024.0, "all_kernels_launches": n_all,
        "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of `bench.py --steps 3 --warmup 2`; "
                  "KB -> bytes, FETCH_SIZE doubled (gfx950 128-B requests tallied at 64 B)",
    }
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
