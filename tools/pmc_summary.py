#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE collected separately).
usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> [out.txt] [note]

FETCH_SIZE / WRITE_SIZE are reported in KiB of 64-byte TCC-EA requests.  On gfx950 FETCH_SIZE counts
a wide (16 B/lane) coalesced read stream at exactly half its bytes (MI355X_MICROARCH.md, HBM section),
so the read column is also shown doubled; WRITE_SIZE is uncalibrated there and shown as reported."""
import csv
import re
import sys
from collections import defaultdict


def load(path, counter):
    acc = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
            acc[name].append(float(row["Counter_Value"]))
    return acc


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    lines = []
    if len(sys.argv) > 4:
        lines.append("# " + sys.argv[4])
    lines.append("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), per dispatch, MB = 1e6 bytes")
    lines.append(f"{'kernel':<64}{'calls':>7}{'fetch_MB':>12}{'fetch_x2_MB':>13}{'write_MB':>12}")
    for name in sorted(fetch, key=lambda k: -sum(fetch[k])):
        if not name.startswith("et::"):
            continue
        fm = sum(fetch[name]) / len(fetch[name]) * 1024 / 1e6
        wm = sum(write.get(name, [0])) / max(len(write.get(name, [0])), 1) * 1024 / 1e6
        lines.append(f"{name[:62]:<64}{len(fetch[name]):>7}{fm:>12.1f}{2 * fm:>13.1f}{wm:>12.1f}")
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(text)
        import json
        js = {}
        for name in fetch:
            if name.startswith("et::"):
                fm = sum(fetch[name]) / len(fetch[name]) * 1024
                wm = sum(write.get(name, [0])) / max(len(write.get(name, [0])), 1) * 1024
                js[name] = dict(read_bytes_corrected=2 * fm, write_bytes=wm, calls=len(fetch[name]))
        json.dump(dict(note=sys.argv[4] if len(sys.argv) > 4 else "", kernels=js),
                  open(sys.argv[3].replace(".txt", ".json"), "w"), indent=1, sort_keys=True)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
