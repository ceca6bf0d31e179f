"""Register / LDS / spill summary of the kernels of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage).

    python scripts/kernel_resources.py fitsnap_amd/csrc/fsnap_syrk.hip [name filter]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"),
           "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    if src.endswith(".cpp"):
        cmd[1:1] = ["-x", "hip"]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    cur = None
    rows = {}
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            rows[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z][^:]*): (\d+)", line)
        if m and cur:
            rows[cur][m.group(1).strip()] = int(m.group(2))
    for name, r in rows.items():
        dem = subprocess.run(["c++filt", name], stdout=subprocess.PIPE, text=True).stdout.strip()
        dem = re.sub(r"\(.*", "", dem)
        if flt and flt not in dem:
            continue
        print(f"{dem:60s} VGPR {r.get('VGPRs', -1):4d} AGPR {r.get('AGPRs', -1):4d} SGPR {r.get('SGPRs', -1):4d} "
              f"spill V {r.get('VGPRs Spill', -1)} S {r.get('SGPRs Spill', -1)} scratch {r.get('ScratchSize [bytes/lane]', -1)} "
              f"LDS {r.get('LDS Size [bytes/block]', -1)} occ {r.get('Occupancy [waves/SIMD]', -1)}")


if __name__ == "__main__":
    main()
