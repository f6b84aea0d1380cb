#!/usr/bin/env python3
"""Register / scratch / LDS / occupancy table of every kernel of one csrc/*.hip file, as the compiler reports it
(-Rpass-analysis=kernel-resource-usage with the Makefile's flags).  *(container)*
usage: tools/kernel_usage.py et_fit.hip [-DFLAG ...] [--grep substring]"""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "eigentrajectory_amd", "csrc")
FLAGS = ("--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 "
         "-fno-slp-vectorize --cuda-device-only -Rpass-analysis=kernel-resource-usage").split()


def main():
    args = sys.argv[1:]
    pat = None
    if "--grep" in args:
        i = args.index("--grep")
        pat = args[i + 1]
        del args[i:i + 2]
    src, extra = args[0], args[1:]
    out = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-c", src, "-o", "/dev/null"], cwd=CSRC,
                         stderr=subprocess.PIPE, text=True).stderr
    rows, cur = [], {}
    for line in out.splitlines():
        m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
        if not m and "error" in line:
            print(line)
        if not m:
            continue
        key, val = m.group(1).split()[0], m.group(2)
        if key == "Function":
            cur = dict(name=val)
            rows.append(cur)
        else:
            cur[key] = val
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), stdout=subprocess.PIPE, text=True).stdout.splitlines()
    print(f"{'vgpr':>5}{'agpr':>5}{'scratch':>8}{'occ':>4}{'lds':>7}  kernel")
    for r, n in zip(rows, names):
        n = re.sub(r"\(.*", "", n).replace("void ", "")
        if pat and pat not in n:
            continue
        print(f"{r.get('VGPRs', '?'):>5}{r.get('AGPRs', '?'):>5}{r.get('ScratchSize', '?'):>8}{r.get('Occupancy', '?'):>4}{r.get('LDS', '?'):>7}  {n}")


if __name__ == "__main__":
    main()
