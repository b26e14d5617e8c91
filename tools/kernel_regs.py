#!/usr/bin/env python3
"""Register / spill / scratch / LDS figures of the kernels in a built library, read from the gfx950 code objects' metadata
(llvm-objdump --offloading + llvm-readelf --notes).  usage: tools/kernel_regs.py [lib.so] [name filter]"""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels(lib):
    out = []
    with tempfile.TemporaryDirectory() as td:
        tmp = os.path.join(td, os.path.basename(lib))
        os.symlink(os.path.abspath(lib), tmp)
        subprocess.run([LLVM + "/llvm-objdump", "--offloading", tmp], capture_output=True, check=True)
        for co in sorted(glob.glob(tmp + ".*gfx950")):
            notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            for blk in notes.split("  - .agpr_count:")[1:]:
                f = {}
                for key in ("name", "vgpr_count", "vgpr_spill_count", "sgpr_count", "sgpr_spill_count", "private_segment_fixed_size",
                            "group_segment_fixed_size"):
                    m = re.search(r"\.%s:\s+(\S+)" % key, blk)
                    f[key] = m.group(1) if m else "?"
                f["agpr"] = blk.split()[0]
                out.append(f)
    return out


def demangle(n):
    return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "optical_rl_gym_amd", "liborlgpu.so")
    filt = sys.argv[2] if len(sys.argv) > 2 else "k_persist"
    print("%-70s %5s %5s %6s %5s %6s %7s" % ("kernel", "vgpr", "agpr", "vspill", "sgpr", "sspill", "scratch"))
    for k in kernels(lib):
        name = demangle(k["name"])
        if filt not in name:
            continue
        name = re.sub(r"^void ", "", name).split("(")[0]
        print("%-70s %5s %5s %6s %5s %6s %7s" % (name, k["vgpr_count"], k["agpr"], k["vgpr_spill_count"], k["sgpr_count"],
                                                 k["sgpr_spill_count"], k["private_segment_fixed_size"]))
