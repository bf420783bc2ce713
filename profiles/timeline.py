#!/usr/bin/env python
"""GPU occupancy of a rocprofv3 --kernel-trace run: union of the kernel intervals over the busiest window,
idle gaps, per-queue busy time and per-kernel totals.  Used to check that gpd_hip_detect_batch keeps the
device busy (two clouds in flight on two streams).

usage: python profiles/timeline.py <results.db> [--last-seconds S]
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    last = float(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[2] == "--last-seconds" else None
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end, queue from kernels order by start").fetchall()
    if not rows:
        print("no kernels")
        return
    t_end = max(r[2] for r in rows)
    if last is not None:
        rows = [r for r in rows if r[1] >= t_end - last * 1e9]
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    # union of intervals
    busy = 0
    gaps = []
    cur_s, cur_e = rows[0][1], rows[0][2]
    for _, s, e, _ in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, cur_e - t0))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    span = t1 - t0
    print("window %.3f ms, %d kernels, device busy (union) %.3f ms = %.1f %%" % (span / 1e6, len(rows), busy / 1e6, 100.0 * busy / span))
    big = sorted(gaps, reverse=True)[:12]
    print("idle gaps: %d, total %.3f ms; > 20 us: %d (%.3f ms)" % (len(gaps), sum(g[0] for g in gaps) / 1e6,
                                                                     sum(1 for g in gaps if g[0] > 20e3),
                                                                     sum(g[0] for g in gaps if g[0] > 20e3) / 1e6))
    print("largest gaps (us @ ms into window):", ", ".join("%.0f@%.2f" % (g[0] / 1e3, g[1] / 1e6) for g in big))
    perq = {}
    for n, s, e, q in rows:
        perq[q] = perq.get(q, 0) + (e - s)
    for q, v in sorted(perq.items()):
        print("  %-10s kernel time %.3f ms" % (q, v / 1e6))
    # overlap: time with >= 2 kernels running
    ev = []
    for _, s, e, _ in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    depth, prev, over = 0, ev[0][0], 0
    for t, d in ev:
        if depth >= 2:
            over += t - prev
        depth += d
        prev = t
    print("time with >= 2 kernels in flight: %.3f ms (%.1f %% of the window)" % (over / 1e6, 100.0 * over / span))
    tot = {}
    for n, s, e, _ in rows:
        k = n.split("(")[0][:60]
        a = tot.setdefault(k, [0, 0])
        a[0] += 1
        a[1] += e - s
    print("%-62s %6s %10s %10s" % ("kernel", "calls", "total_ms", "avg_us"))
    for k, (n, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:20]:
        print("%-62s %6d %10.3f %10.2f" % (k, n, t / 1e6, t / n / 1e3))


if __name__ == "__main__":
    main()
