#!/usr/bin/env python3
"""Register / LDS / scratch use of the kernels in libhmx.so: `[HMX_LIB_PATH=variant.so] python tools/kernel_regs.py [regex]` (reads the AMDGPU metadata notes of every
gfx950 code object in the library; vgpr = .vgpr_count of the metadata = architectural + accumulation registers TOGETHER on gfx950's unified file: waves per SIMD = 512 / vgpr -- confirmed by SQ_WAVE_CYCLES in round 6: a kernel at 240 (40 of them AGPRs) runs two waves per SIMD, at 264 one)."""
import os
import pathlib
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_isa_shape as t  # noqa: E402

pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else ".")
t.LIB = os.environ.get("HMX_LIB_PATH", t.LIB)  # a build variant (tools/variant.sh)
with tempfile.TemporaryDirectory() as d:
    rows = []
    for co in t.code_objects(pathlib.Path(d)):
        notes = subprocess.run([os.path.join(t.LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            f = dict(re.findall(r"\.(\w+):\s+(\S+)", blk))
            f["agpr_count"] = blk.split()[0]
            name = subprocess.run(["c++filt", f["name"]], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(.*", "", name).replace("void ", "")
            if pat.search(name):
                v, a = int(f["vgpr_count"]), int(f["agpr_count"])
                rows.append((name, v, a, int(f["sgpr_count"]), int(f["group_segment_fixed_size"]), int(f["private_segment_fixed_size"]), min(8, 512 // max(1, (v + 7) // 8 * 8))))
    print("%-52s %5s %5s %5s %7s %7s %6s" % ("kernel", "vgpr", "agpr", "sgpr", "lds", "scratch", "waves"))
    for r in sorted(rows):
        print("%-52s %5d %5d %5d %7d %7d %6d" % r)
