#!/usr/bin/env python3
"""Instruction mix of the largest basic block of one kernel in the device ISA.

usage: isa_stats.py <kernel-name-substring> [extra hipcc flags ...]
Compiles tricolour_amd/csrc/tricolour_amd.hip to gfx950 assembly with the
library's flags (tricolour_amd._lib.HIPCC_FLAGS) and prints register use plus
an opcode histogram of the kernel's largest basic block (the unrolled hot loop).
"""
import collections, os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tricolour_amd import _lib

def main():
    name, extra = sys.argv[1], sys.argv[2:]
    flags = [f for f in _lib.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    if "--no-max-ilp" in extra:      # drop "-mllvm -amdgpu-sched-strategy=max-ilp"
        extra.remove("--no-max-ilp")
        i = flags.index("-amdgpu-sched-strategy=max-ilp")
        del flags[i - 1:i + 1]
    out = os.path.join(tempfile.gettempdir(), "tri_isa.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + extra + ["-S", "--cuda-device-only", "-o", out] + _lib.SOURCES,
                          stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(name), l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    for l in lines[end:end + 80]:
        if re.search(r"; (NumVgprs|NumSgprs|NumAgprs|Occupancy|ScratchSize|codeLenInByte)", l):
            print(l.strip())
    blocks, cur = collections.OrderedDict(), "entry"
    blocks[cur] = []
    for l in lines[start:end]:
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            cur = m.group(1)
            blocks[cur] = []
        elif re.match(r"^\t[a-z]", l) and not l.startswith("\t."):
            blocks[cur].append(l.split()[0])
    for big in sorted(blocks, key=lambda b: -len(blocks[b]))[:int(os.environ.get("ISA_BLOCKS", "1"))]:
        ops = blocks[big]
        print("block", big, len(ops), "instructions: VALU", sum(o.startswith("v_") for o in ops),
              "SALU", sum(o.startswith("s_") for o in ops), "MEM", sum(o.startswith(("global", "buffer", "ds_")) for o in ops),
              "spill", sum(o.startswith(("v_readlane", "v_writelane")) for o in ops))
        for op, n in collections.Counter(ops).most_common(int(os.environ.get("ISA_TOP", "30"))):
            print("%6d %s" % (n, op))

if __name__ == "__main__":
    main()
