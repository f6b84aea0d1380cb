#!/usr/bin/env python3
"""Summarise a rocprofv3 result (rocpd SQLite db written by `rocprofv3 --kernel-trace --stats`)
into a small text table for profiles/.  usage: rocprof_summary.py <results.db> [out.txt] [note]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)          # drop the argument list
    name = name.replace("void ", "")
    return name if len(name) <= 90 else name[:87] + "..."


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    lines = []
    if len(sys.argv) > 3:
        lines.append("# " + sys.argv[3])
    lines.append("# source: rocprofv3 --kernel-trace --stats (durations in microseconds)")
    lines.append(f"{'kernel':<92}{'calls':>8}{'total_us':>14}{'avg_us':>12}{'pct':>8}")
    for name, calls, total, avg, pct in rows:
        lines.append(f"{short(name):<92}{calls:>8}{total:>14.1f}{avg:>12.2f}{pct:>8.2f}")
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
