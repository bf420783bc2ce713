#!/usr/bin/env python
"""Runs a command in its own session and kills exactly that session when its resident host memory (sum of VmRSS over the
session's processes) passes a limit or a wall-clock limit expires.  Round 5 lost three GPU boxes to one unbounded host-side
leg of bench.py; every collection script of round 6 runs its commands under this guard.

  python profiles/memguard.py --rss-gb 24 --seconds 600 -- python bench.py --config 4
exit code: the command's, 137 when the guard killed it (the reason goes to stderr).
"""
import argparse
import os
import signal
import subprocess
import sys
import time


def session_rss_kb(sid):
    total = 0
    for pid in os.listdir("/proc"):
        if not pid.isdigit():
            continue
        try:
            with open("/proc/%s/stat" % pid) as f:
                st = f.read()
            # field 6 (after the parenthesised command name) is the session id
            if int(st[st.rindex(")") + 2:].split()[3]) != sid:
                continue
            with open("/proc/%s/status" % pid) as f:
                for line in f:
                    if line.startswith("VmRSS:"):
                        total += int(line.split()[1])
                        break
        except (OSError, ValueError, IndexError):
            continue
    return total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rss-gb", type=float, default=24.0)
    ap.add_argument("--seconds", type=float, default=900.0)
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == "--" else a.cmd
    p = subprocess.Popen(cmd, start_new_session=True)
    t0 = time.time()
    peak = 0
    why = None
    while p.poll() is None:
        time.sleep(0.25)
        rss = session_rss_kb(p.pid)
        peak = max(peak, rss)
        if rss > a.rss_gb * 1048576.0:
            why = "resident host memory %.1f GB > %.1f GB" % (rss / 1048576.0, a.rss_gb)
        elif time.time() - t0 > a.seconds:
            why = "wall clock > %.0f s" % a.seconds
        if why:
            try:
                os.killpg(p.pid, signal.SIGKILL)  # its own session: exactly the processes started here
            except ProcessLookupError:
                pass
            p.wait()
            sys.stderr.write("memguard: killed (%s): %s\n" % (why, " ".join(cmd)))
            sys.exit(137)
    sys.stderr.write("memguard: peak resident host memory %.2f GB, %.1f s: %s\n" % (peak / 1048576.0, time.time() - t0, " ".join(cmd)[:120]))
    sys.exit(p.returncode)


if __name__ == "__main__":
    main()
