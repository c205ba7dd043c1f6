#!/usr/bin/env python3
"""VGPR / spill / scratch / LDS of the kernels in a built object (reads the code object's metadata notes).
  python tools/kernel_resources.py gemm [name-filter]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin/"
obj = os.path.join(ROOT, "whisper-char-alignment_amd", "build", sys.argv[1] + ".o")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
subprocess.check_call([LLVM + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, "/tmp/_k.fatbin"])
subprocess.check_call([LLVM + "clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=/tmp/_k.fatbin",
                       "--output=/tmp/_k.co", "--unbundle"])
notes = subprocess.run([LLVM + "llvm-readelf", "--notes", "/tmp/_k.co"], capture_output=True, text=True).stdout
for ent in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, ent) or [None, "?"])[1]  # noqa: E731
    name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
    if flt in name:
        print("%-110s vgpr %s spill %s sgpr %s sspill %s scratch %s lds %s" % (name[:110], g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_count"),
                                                                              g("sgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
