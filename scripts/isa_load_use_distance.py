#!/usr/bin/env python
"""
Load-to-use distances in a kernel's compiled ISA: for every `global_load_dwordx4` / `ds_read_b128` of a kernel, the number
of MFMAs between the load and the first MFMA that reads its destination registers.  A distance of a few MFMAs means the
wave waits for (most of) the memory round trip right there; only another wave of the SIMD covers it.

    hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Itextocvp_amd/csrc -c textocvp_amd/csrc/conv_f16x3.hip \
          -o /tmp/conv_f16x3.o -save-temps=obj
    python scripts/isa_load_use_distance.py /tmp/conv_f16x3-hip-amdgcn-amd-amdhsa-gfx950.s conv5x5_dec_f16x3_kernelILi0ELb1ELb0E

Found in round 4 (profiles/r04_conv.md): the decoder conv consumed 16 of a pass's 96 weight fragments in the very next MFMA.
"""
import re
import sys


def distances(ins, pat):
    res = []
    for i, l in enumerate(ins):
        m = re.match(pat + r" (v\[\d+:\d+\])", l)
        if not m:
            continue
        reg, n = m.group(1), 0
        for j in range(i + 1, min(len(ins), i + 800)):
            if ins[j].startswith("v_mfma"):
                ops = [o.strip() for o in ins[j].split(",")]
                if reg in ops[1:3]:
                    res.append(n)
                    break
                n += 1
    return sorted(res)


def main():
    path, needle = sys.argv[1], sys.argv[2]
    src = open(path).read().split("\n")
    starts = [i for i, l in enumerate(src) if needle in l and l.rstrip().split(" ")[0].endswith(":") and l.startswith("_Z")]
    for start in starts:
        end = next(i for i in range(start, len(src)) if "s_endpgm" in src[i])
        ins = [l.strip() for l in src[start:end] if l.strip() and not l.strip().startswith((";", "."))]
        g, d = distances(ins, "global_load_dwordx4"), distances(ins, "ds_read_b128")
        mf = sum(1 for l in ins if l.startswith("v_mfma"))
        print(src[start].split(":")[0])
        print(f"  MFMAs {mf}")
        for name, v in (("global_load_dwordx4 -> MFMA", g), ("ds_read_b128 -> MFMA", d)):
            if v:
                print(f"  {name}: n {len(v)}, smallest {v[:12]}, median {v[len(v) // 2]}")


if __name__ == "__main__":
    main()
