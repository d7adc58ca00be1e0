#!/usr/bin/env python3
"""Where do the SGPR spills of a kernel sit?  Compiles csrc/gemm_select.hip to gfx950 assembly (device only) and lists, per
basic block of one kernel, the MFMAs, barriers and the v_writelane / v_readlane instructions the compiler uses to spill and
reload scalar registers, with the loop depth the assembler comments give.  CPU only (hipcc cross-compiles).

    python scripts/spill_report.py [mangled-kernel-substring]      default: the product instantiation of gemm_tile_kernel
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "image-search-engine-for-historical-research_amd", "csrc", "gemm_select.hip")
want = sys.argv[1] if len(sys.argv) > 1 else "gemm_tile_kernelILb0ELi0ELb1ELb0ELi3E"

with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "k.s")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-S", "--cuda-device-only",
                        "-Wno-unused-value", "-Rpass-analysis=kernel-resource-usage", "-o", out, SRC],
                       capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr[-3000:])
    usage = r.stderr.split("\n")
    lines = open(out).read().split("\n")

for i, l in enumerate(usage):
    if "Function Name" in l and want in l:
        print("\n".join(x.split("remark:")[1].replace("[-Rpass-analysis=kernel-resource-usage]", "").rstrip()
                        for x in usage[i:i + 11] if "remark:" in x))
        break
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and want in l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
body = lines[start:end]
blocks, cur = collections.OrderedDict(), "(entry)"
blocks[cur] = collections.Counter()
for l in body:
    m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?", l)
    if m:
        cur = m.group(1) + ("  " + m.group(2).strip() if m.group(2) else "")
        blocks[cur] = collections.Counter()
        continue
    for key in ("v_mfma", "s_barrier", "v_writelane_b32", "v_readlane_b32", "scratch_", "s_waitcnt vmcnt(0)"):
        if key in l:
            blocks[cur][key] += 1
print("%d instructions-lines, %d blocks" % (len(body), len(blocks)))
print("%-62s %5s %5s %9s %8s" % ("block", "mfma", "barr", "writelane", "readlane"))
for name, c in blocks.items():
    if c["v_mfma"] or c["v_writelane_b32"] or c["v_readlane_b32"] or c["scratch_"]:
        print("%-62s %5d %5d %9d %8d%s" % (name[:62], c["v_mfma"], c["s_barrier"], c["v_writelane_b32"], c["v_readlane_b32"],
                                           "  scratch %d" % c["scratch_"] if c["scratch_"] else ""))
tot = collections.Counter()
for name, c in blocks.items():
    depth = int(re.search(r"Depth=(\d+)", name).group(1)) if "Depth=" in name else 0
    tot[depth] += c["v_writelane_b32"] + c["v_readlane_b32"]
print("spill / reload instructions by loop depth (0 = outside loops, 1 = per tile, 2 = per K-slice):", dict(tot))
