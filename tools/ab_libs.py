#!/usr/bin/env python3
"""Same-box A/B of library variants (tools/build_variant.sh) on the bench step: runs bench.py alternately with each
library, prints the stage times.   python tools/ab_libs.py base xcdmap [rounds]"""
import json
import os
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names = [a for a in sys.argv[1:] if not a.isdigit()]
rounds = int([a for a in sys.argv[1:] if a.isdigit()][0]) if any(a.isdigit() for a in sys.argv[1:]) else 2
for r in range(rounds):
    for name in names:
        env = dict(os.environ)
        if name != "base":
            env["ET_LIBETAMD"] = os.path.join(R, "eigentrajectory_amd", "variants", f"libetamd_{name}.so")
        out = subprocess.run([sys.executable, os.path.join(R, "bench.py"), "--steps", "20", "--warmup", "5", "--no-extras",
                              "--no-cpu-baseline"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        d = json.loads(out)
        st = d["stages"]
        print(f"round {r} {name:10s} step {d['ms_per_step']:.3f}  " +
              "  ".join(f"{k} {v['ms']:.4f}" + (f" ({v['frac_of_peak']:.3f})" if 'frac_of_peak' in v else "") for k, v in st.items()),
              flush=True)
