#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel whose name contains <substring>:
kernel_resources.py <substring> [extra hipcc flags ...]"""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tricolour_amd import _lib
pat, extra = sys.argv[1], sys.argv[2:]
flags = [f for f in _lib.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
if "--no-max-ilp" in extra:      # drop "-mllvm -amdgpu-sched-strategy=max-ilp"
    extra.remove("--no-max-ilp")
    i = flags.index("-amdgpu-sched-strategy=max-ilp")
    del flags[i - 1:i + 1]
out = os.path.join(tempfile.gettempdir(), "tri_res.s")
subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + extra + ["-S", "--cuda-device-only", "-o", out] + _lib.SOURCES, stderr=subprocess.DEVNULL)
txt = open(out).read()
names = sorted(set(re.findall(r"^\t\.amdhsa_kernel (\S+)", txt, re.M)))
for name in names:
    if pat not in name:
        continue
    def g(k):
        r = re.search(r"\.set %s\.%s, (\S+)" % (re.escape(name), k), txt)
        return r.group(1) if r else "?"
    j = txt.index(".Lfunc_end", txt.index("\n" + name + ":"))
    tail = txt[j:j + 6000]
    def c(k):
        r = re.search(r"; %s[:=]? *=? *(\d+)" % k, tail)
        return r.group(1) if r else "?"
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    print("%-46s vgpr %3s agpr %3s scratch %5s occ %s code %s" % (dem[-46:], g("num_vgpr"), g("num_agpr"), g("private_seg_size"), c("Occupancy"), c("codeLenInByte")))
