#!/usr/bin/env python3
"""Register / spill / LDS figures of every kernel in ptcore.hip from a device-only compile (no GPU needed).
   python tools/kernel_regs.py [extra hipcc flags]      ISA is left in $TMPDIR/ptcore_isa/ptcore.s"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(os.environ.get("TMPDIR", "/tmp"), "ptcore_isa")
os.makedirs(out, exist_ok=True)
s = os.path.join(out, "ptcore.s")
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-S",
                "--cuda-device-only", "-I" + os.path.join(ROOT, "include"), *sys.argv[1:],
                os.path.join(ROOT, "path_trace_golang_amd", "csrc", "ptcore.hip"), "-o", s],
               check=True, stderr=subprocess.DEVNULL)
text = open(s).read()
meta = text[text.index(".amdgpu_metadata"):]
for blk in meta.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name)
    v = int(g("vgpr_count"))
    waves = min(8, 512 // max(-(-v // 8) * 8, 8))  # VGPRs are allocated in blocks of 8
    print("%-48s vgpr %3d (<=%d waves/SIMD) sgpr %3s spill v%s s%s lds %6s scratch %s" % (
        name[-48:], v, waves, g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"),
        g("group_segment_fixed_size"), g("private_segment_fixed_size")))
print("ISA:", s)
