#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS use from the saved ISA (python -m igate4xsoftphonedsp_amd.build --asm)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASM_DIR = os.path.join(ROOT, "igate4xsoftphonedsp_amd", "_asm")
ASM = os.path.join(ASM_DIR, "igdsp_k_meter-hip-amdgcn-amd-amdhsa-gfx950.s")      # one of the translation units: staleness check


def asm_files():
    import glob
    return sorted(glob.glob(os.path.join(ASM_DIR, "igdsp_k_*-hip-amdgcn-amd-amdhsa-gfx950.s")))


def resources(paths=None):
    out = []
    for path in (paths or asm_files()):
        out += _resources_of(path)
    return out


def _resources_of(path):
    s = open(path).read()
    meta = s[s.index("amdhsa.kernels:"):]
    out = []
    for blk in meta.split("  - .agpr_count:")[1:]:
        g = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk).group(1)
        out.append({"name": g("name"), "vgpr": int(g("vgpr_count")), "sgpr": int(g("sgpr_count")), "spill": int(g("vgpr_spill_count")),
                    "sgpr_spill": int(g("sgpr_spill_count")), "scratch": int(g("private_segment_fixed_size")), "lds": int(g("group_segment_fixed_size"))})
    names = subprocess.run(["c++filt"] + [r["name"] for r in out], capture_output=True, text=True).stdout.split("\n")
    for r, d in zip(out, names):
        r["demangled"] = d.split("(")[0]
        r["file"] = os.path.basename(path).split("-")[0]
    return out

if __name__ == "__main__":
    for r in resources():
        if len(sys.argv) < 2 or sys.argv[1] in r["demangled"]:
            print(f"{r['demangled']:<60} vgpr {r['vgpr']:>3} sgpr {r['sgpr']:>3} spill {r['spill']} scratch {r['scratch']:>3} lds {r['lds']}")
