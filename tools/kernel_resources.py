#!/usr/bin/env python3
"""Register / scratch / occupancy table of every kernel in libmisti_hip.so (CPU only: hipcc's resource remarks).

    python tools/kernel_resources.py > profiles/rNN_kernel_resources.txt"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from misti_amd import build
    cmd = [build.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-ffp-contract=on", "--cuda-device-only", "-c",
           "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null"]
    rows = {}
    for src in ("misti_kernels.hip", "misti_nm.hip"):
        out = subprocess.run(cmd + ["-x", "hip", os.path.join(build.CSRC, src)], capture_output=True, text=True).stderr
        cur = None
        for line in out.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                cur = m.group(1)
                rows[cur] = {}
                continue
            for key, tag in (("VGPRs:", "vgpr"), ("AGPRs:", "agpr"), ("ScratchSize", "scratch"), ("Occupancy", "occ"), ("SGPRs Spill", "sspill"),
                             ("VGPRs Spill", "vspill"), ("LDS Size", "lds")):
                if key in line and cur:
                    rows[cur][tag] = line.split(":")[-1].split("[")[0].strip()
    print("%-64s %5s %5s %8s %4s %11s %11s" % ("kernel", "VGPR", "AGPR", "scratch", "occ", "SGPR spills", "VGPR spills"))
    for k, v in rows.items():
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")
        print("%-64s %5s %5s %8s %4s %11s %11s" % (name[:64], v.get("vgpr"), v.get("agpr"), v.get("scratch"), v.get("occ"), v.get("sspill"), v.get("vspill")))


if __name__ == "__main__":
    main()
