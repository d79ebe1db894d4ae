#!/usr/bin/env python3
"""Instruction mix of one kernel's main loop from the saved ISA (build.py --asm).
usage: tools/asm_mix.py <kernel-substring> [--loop]   (--loop: only the largest loop body)"""
import collections
import re
import sys

import glob

ASMS = sorted(glob.glob("igate4xsoftphonedsp_amd/_asm/igdsp_k_*-hip-amdgcn-amd-amdhsa-gfx950.s"))


def main():
    key = sys.argv[1]
    for path in ASMS:                                 # the translation unit that defines the kernel
        lines = open(path).read().split("\n")
        if any(re.match(r"^_Z\w*:", l) and key in l for l in lines):
            break
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and key in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start + 1:end]
    if "--loop" in sys.argv:
        # largest backward branch span
        labels = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"^\.LBB\w+:", l)}
        best = (0, 0, 0)
        for i, l in enumerate(body):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\w+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                span = i - labels[m.group(1)]
                if span > best[0]:
                    best = (span, labels[m.group(1)], i)
        body = body[best[1]:best[2] + 1]
    cnt = collections.Counter()
    for l in body:
        l = l.strip()
        if not l or l[0] in ";." or l.endswith(":"):
            continue
        cnt[l.split()[0]] += 1
    cls = collections.Counter()
    for op, c in cnt.items():
        k = ("valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else "salu" if op.startswith("s_")
             else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
        cls[k] += c
    print(lines[start].split(":")[0], "instructions:", sum(cnt.values()), dict(cls))
    for op, c in cnt.most_common(40):
        print(f"  {c:6d}  {op}")


if __name__ == "__main__":
    main()
