#!/usr/bin/env python
"""Turns the rocprofv3 rocpd databases written by profiles/run_profile.sh into the text
summary committed under profiles/ (per-kernel count / total / average duration, and the
FETCH_SIZE / WRITE_SIZE PMC sums per kernel).

usage: python profiles/summarize.py gpurun_out/prof_<tag> > profiles/<tag>_rocprof_summary.txt
"""
import os
import sqlite3
import sys


def kernel_stats(db):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     "from kernels group by name order by sum(end-start) desc").fetchall()
    return rows


def counter_stats(db):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    rows = c.execute("select %s, counter_name, count(*), sum(value), avg(value) from counters_collection "
                     "group by %s, counter_name order by sum(value) desc" % (name_col, name_col)).fetchall()
    return rows


def main():
    d = sys.argv[1]
    print("# rocprofv3 summary of %s" % d)
    st = os.path.join(d, "stats", "bench_results.db")
    if os.path.exists(st):
        print("\n## kernel-trace --stats  (bench.py --steps 10 --warmup 2 --batch-clouds 0: the 12 replays of the 5000-candidate list plus")
        print("##   the first passes and the gpd_hip_detect calls on the 6077 candidates of the whole sample set; `timed_us` = average")
        print("##   over the LAST 10 launches, i.e. the timed steps, which is what bench.py's HIP events measure)")
        print("%-70s %6s %12s %12s %12s %12s %12s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "timed_us"))
        c = sqlite3.connect(st)
        timed = {}
        for (name,) in c.execute("select distinct name from kernels").fetchall():
            d10 = [r[0] for r in c.execute("select end-start from kernels where name=? order by start desc limit 10", (name,)).fetchall()]
            timed[name] = sum(d10) / len(d10)
        tot = 0.0
        rows = kernel_stats(st)
        for name, n, total, avg, mn, mx in rows:
            tot += total
        for name, n, total, avg, mn, mx in rows:
            print("%-70s %6d %12.1f %12.2f %12.2f %12.2f %12.2f  %5.1f%%" % (name[:70], n, total / 1e3, avg / 1e3, mn / 1e3, mx / 1e3,
                                                                             timed[name] / 1e3, 100.0 * total / tot))
    for sub, label in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        db = os.path.join(d, sub, "bench_results.db")
        if not os.path.exists(db):
            continue
        print("\n## --pmc %s  (bench.py --steps 3 --warmup 1 => 5 launches; raw counter units = KiB as rocprofv3 reports them;" % label)
        print("##   per MI355X_MICROARCH.md §HBM FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950)")
        print("%-70s %-12s %6s %16s %16s" % ("kernel", "counter", "calls", "sum", "avg_per_launch"))
        for name, cn, n, s, a in counter_stats(db):
            print("%-70s %-12s %6d %16.1f %16.1f" % (str(name)[:70], cn, n, s, a))


def traffic_json(d, out):
    """Per-launch HBM traffic of every kernel from the two PMC passes.  FETCH_SIZE / WRITE_SIZE are
    in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads
    (MI355X_MICROARCH.md §HBM), so read bytes = 2 * FETCH_SIZE * 1024.  Calibration in this repo:
    conv1 reads exactly n*C*3600 image bytes with 16-B lanes and FETCH_SIZE reports half of that."""
    import json
    res = {}
    for sub, key in (("pmc_fetch", "fetch_KiB_raw"), ("pmc_write", "write_KiB")):
        db = os.path.join(d, sub, "bench_results.db")
        if not os.path.exists(db):
            continue
        for name, cn, n, s, a in counter_stats(db):
            res.setdefault(str(name), {})[key] = a
    for k, v in res.items():
        f, w = v.get("fetch_KiB_raw", 0.0), v.get("write_KiB", 0.0)
        v["read_bytes_corrected"] = 2.0 * f * 1024.0
        v["write_bytes"] = w * 1024.0
        v["hbm_bytes_per_launch"] = v["read_bytes_corrected"] + v["write_bytes"]
    import os as _os
    import sys as _sys
    _sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
    import bench  # the SHA-1 of the kernel sources this was measured on: bench.py drops the numbers when they change
    json.dump({"source": d, "source_hashes": bench.source_hashes(), "kernels": res}, open(out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[2] == "--traffic":
        traffic_json(sys.argv[1], sys.argv[3])
    else:
        main()
