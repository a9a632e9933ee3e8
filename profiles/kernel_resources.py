#!/usr/bin/env python3
"""Static per-kernel resource table (no GPU involved): compiles every .hip of the product with
-Rpass-analysis=kernel-resource-usage and tabulates VGPRs / SGPRs / private memory / spills / LDS / occupancy.

    python profiles/kernel_resources.py > profiles/rNN_kernel_resources.txt
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mpeg-pcc-tmc2_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
         "-I" + CSRC, "-Rpass-analysis=kernel-resource-usage"]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    res = []
    for line in out.splitlines():
        line = re.sub(r"\(anonymous namespace\)::", "", line)
        line = re.sub(r"^(void )?tmc2::", "", line)
        line = re.sub(r"^void ", "", line)
        depth, cut = 0, len(line)
        for i, ch in enumerate(line):                      # drop the argument list, keep template arguments
            if ch == "<":
                depth += 1
            elif ch == ">":
                depth -= 1
            elif ch == "(" and depth == 0:
                cut = i
                break
        res.append(line[:cut])
    return res


def main():
    rows = []
    for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
        with tempfile.NamedTemporaryFile(suffix=".o") as obj:
            err = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-c", src, "-o", obj.name], capture_output=True, text=True).stderr
        cur = None
        for line in err.splitlines():
            m = re.search(r"remark: Function Name: (\S+)", line)
            if m:
                cur = {"src": os.path.basename(src), "name": m.group(1)}
                rows.append(cur)
                continue
            m = re.search(r"remark:\s+([A-Za-z /\[\]]+): (\S+) \[-Rpass", line)
            if m and cur is not None:
                cur[m.group(1).strip()] = m.group(2)
    names = demangle([r["name"] for r in rows])
    print("# per-kernel resource usage, gfx950, hipcc -O3 -Rpass-analysis=kernel-resource-usage (static; profiles/kernel_resources.py)")
    print("# occ = waves per SIMD the register / LDS budget allows (8 = full); scratch = private memory per lane in bytes (local arrays)")
    print("# spill = VGPRs spilled; LDS = static bytes per workgroup (dynamic LDS is sized by the host at launch)")
    print(f"{'source':<22} {'kernel':<52} {'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'scratch':>8} {'spill':>6} {'LDS':>7} {'occ':>4}")
    for r, n in zip(rows, names):
        print(f"{r['src']:<22} {n[:52]:<52} {r.get('VGPRs', '?'):>5} {r.get('AGPRs', '?'):>5} {r.get('TotalSGPRs', '?'):>5} "
              f"{r.get('ScratchSize [bytes/lane]', '?'):>8} {r.get('VGPRs Spill', '?'):>6} {r.get('LDS Size [bytes/block]', '?'):>7} "
              f"{r.get('Occupancy [waves/SIMD]', '?'):>4}")
    print(f"# {len(rows)} kernels; VGPR spills: {sum(int(r.get('VGPRs Spill', 0)) for r in rows)}")


if __name__ == "__main__":
    sys.exit(main())
