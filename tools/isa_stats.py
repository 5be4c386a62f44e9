"""Kernel register / spill / LDS figures out of an AMDGPU assembly listing (hipcc -S --cuda-device-only).
usage: python tools/isa_stats.py file.s [substring]"""
import re
import subprocess
import sys

text = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"- \.agpr_count.*?\.wavefront_size", text, re.S):
    blk = m.group(0)
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if want not in name:
        continue
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], stdout=subprocess.PIPE).stdout.decode().strip()
    except OSError:
        pass
    g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)
    print("%s\n    vgpr %s agpr %s sgpr %s  vgpr_spill %s sgpr_spill %s  lds %s scratch %s" % (
        name.split("(")[0], g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"),
        g("group_segment_fixed_size"), g("private_segment_fixed_size")))
