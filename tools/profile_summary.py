#!/usr/bin/env python3
"""Text / json summaries of one tools/profile_round.sh run (rocprofv3 csv -> what is committed under profiles/).

usage: profile_summary.py <gpurun_out/prof_TAG> <TAG>

  <TAG>_bench_kernel_stats.txt   per-kernel calls / total / average duration (rocprofv3 --kernel-trace --stats)
  <TAG>_pmc_hbm_traffic.txt|json per-kernel HBM bytes per launch: FETCH_SIZE (doubled: gfx950 counts a 16 B/lane
                                 read stream at half its bytes, MI355X_MICROARCH.md HBM section) + WRITE_SIZE
  <TAG>_sq_breakdown.txt         SQ counters per launch of every et:: kernel (means over launches; the first launch of
                                 a kernel name is dropped when there are more than 4: warm-up / fallback path)
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def kname(raw):
    return re.sub(r"\(.*", "", raw).replace("void ", "")


def find(d, pattern):
    hits = glob.glob(os.path.join(d, "**", pattern), recursive=True)
    return hits[0] if hits else None


def counters(d):
    path = find(d, "*counter_collection.csv")
    acc = defaultdict(lambda: defaultdict(list))
    if not path:
        return acc
    with open(path) as f:
        for row in csv.DictReader(f):
            acc[kname(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc


def main():
    out, tag = sys.argv[1], sys.argv[2]
    # ---- kernel stats
    path = find(os.path.join(out, "stats"), "*kernel_stats.csv")
    if path:
        rows = list(csv.DictReader(open(path)))
        lines = ["# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 1   (N=1e+07)",
                 f"{'kernel':<72}{'calls':>8}{'total_ms':>12}{'avg_us':>12}{'pct':>8}"]
        for r in rows:
            lines.append(f"{kname(r['Name'])[:70]:<72}{int(r['Calls']):>8}{float(r['TotalDurationNs']) / 1e6:>12.3f}"
                         f"{float(r['AverageNs']) / 1e3:>12.2f}{float(r['Percentage']):>8.2f}")
        open(os.path.join(out, f"{tag}_bench_kernel_stats.txt"), "w").write("\n".join(lines) + "\n")
    # ---- HBM traffic
    fetch, write = counters(os.path.join(out, "fetch")), counters(os.path.join(out, "write"))
    lines = ["# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --no-cpu-baseline "
             "--no-extras --steps 3 --warmup 1   (N=1e+07); per launch, MB = 1e6 bytes",
             f"{'kernel':<72}{'calls':>7}{'fetch_MB':>11}{'fetch_x2_MB':>13}{'write_MB':>11}"]
    js = {}
    for name in sorted(fetch, key=lambda k: -sum(fetch[k].get("FETCH_SIZE", [0]))):
        if not name.startswith("et::"):
            continue
        f = fetch[name].get("FETCH_SIZE", [0.0])
        w = write.get(name, {}).get("WRITE_SIZE", [0.0])
        fm, wm = sum(f) / len(f) * 1024, sum(w) / len(w) * 1024
        lines.append(f"{name[:70]:<72}{len(f):>7}{fm / 1e6:>11.1f}{2 * fm / 1e6:>13.1f}{wm / 1e6:>11.1f}")
        js[name] = dict(read_bytes_corrected=2 * fm, write_bytes=wm, calls=len(f))
    open(os.path.join(out, f"{tag}_pmc_hbm_traffic.txt"), "w").write("\n".join(lines) + "\n")
    json.dump(dict(note="bench.py N=1e7", kernels=js), open(os.path.join(out, f"{tag}_pmc_hbm_traffic.json"), "w"),
              indent=1, sort_keys=True)
    # ---- SQ breakdown
    merged = defaultdict(dict)
    for i in (1, 2, 3):
        for name, cs in counters(os.path.join(out, f"sq{i}")).items():
            if not name.startswith("et::"):
                continue
            for c, vals in cs.items():
                vals = vals[1:] if len(vals) > 4 else vals
                merged[name][c] = (sum(vals) / len(vals), len(vals))
    lines = ["# rocprofv3 --pmc <SQ counters> (three passes) -- python bench.py --no-cpu-baseline --no-extras --steps 3 "
             "--warmup 1   (N=1e+07); mean per launch over the whole chip",
             "# derived: valu_busy = SQ_ACTIVE_INST_VALU * 4 / SQ_BUSY_CYCLES / (SIMDs per SE ...) is NOT attempted; "
             "read the ratios: ACTIVE_INST_VALU / WAVE_CYCLES (share of a wave's life issuing VALU), "
             "INSTS_VALU / WAVES (instructions per wave), MFMA_BUSY / BUSY_CYCLES"]
    for name in sorted(merged, key=lambda k: -merged[k].get("SQ_WAVE_CYCLES", (0, 0))[0]):
        cs = merged[name]
        lines.append(name)
        for c in sorted(cs):
            lines.append(f"    {c:<32}{cs[c][0]:>16.4g}   (n={cs[c][1]})")
        g = lambda c: cs.get(c, (0.0, 0))[0]
        if g("SQ_WAVE_CYCLES") and g("SQ_WAVES"):
            lines.append(f"    -> VALU instructions / wave        {g('SQ_INSTS_VALU') / g('SQ_WAVES'):>12.1f}")
            lines.append(f"    -> ACTIVE_INST_VALU / WAVE_CYCLES   {g('SQ_ACTIVE_INST_VALU') / g('SQ_WAVE_CYCLES'):>12.3f}")
            lines.append(f"    -> WAIT_INST_ANY / WAVE_CYCLES      {g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'):>12.3f}")
            if g("SQ_BUSY_CYCLES"):
                lines.append(f"    -> MFMA_BUSY / BUSY_CYCLES          {g('SQ_VALU_MFMA_BUSY_CYCLES') / g('SQ_BUSY_CYCLES'):>12.3f}")
    open(os.path.join(out, f"{tag}_sq_breakdown.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:60]))


if __name__ == "__main__":
    main()
