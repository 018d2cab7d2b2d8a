#!/usr/bin/env python3
"""Per-kernel resource metadata of the gfx950 code object inside libfiat_amd.so (build container or GPU box; needs only
the LLVM tools shipped with ROCm): name, VGPRs, AGPRs, SGPR / VGPR spills, private (scratch) bytes per lane, LDS.
`python tools/codeobject_report.py [--scratch]`; `kernels()` is used by tests/test_codeobject.py."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("ROCM_LLVM", "/opt/rocm/lib/llvm/bin")
LIB = os.path.join(ROOT, "fiat_amd", "csrc", "libfiat_amd.so")
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def kernels(lib=LIB):
    """[{name, vgpr, agpr, sgpr_spill, vgpr_spill, scratch, lds}] and the list of bundle targets in the library."""
    with tempfile.TemporaryDirectory() as tmp:
        fat, dev = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", lib], check=True,
                       capture_output=True)
        listing = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--list", "--type=o", f"--input={fat}"],
                                 check=True, capture_output=True, text=True).stdout.split()
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                        f"--targets={TARGET}", f"--output={dev}"], check=True, capture_output=True)
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", dev], check=True, capture_output=True,
                               text=True).stdout
    out = []
    for block in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
        def num(key):
            m = re.search(rf"\.{key}:\s+(\d+)", block)
            return int(m.group(1)) if m else 0
        out.append({"name": re.search(r"\.name:\s+(\S+)", block).group(1), "agpr": int(block.split()[0]),
                    "vgpr": num("vgpr_count"), "sgpr_spill": num("sgpr_spill_count"), "vgpr_spill": num("vgpr_spill_count"),
                    "scratch": num("private_segment_fixed_size"), "lds": num("group_segment_fixed_size")})
    return out, listing


def demangled(names):
    for tool in (os.path.join(LLVM, "llvm-cxxfilt"), "c++filt"):
        try:
            res = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True)
            if res.returncode == 0:
                return res.stdout.split("\n")
        except FileNotFoundError:
            pass
    return names


if __name__ == "__main__":
    ks, targets = kernels()
    print(f"{len(ks)} kernels, bundle targets: {targets}")
    only_scratch = "--scratch" in sys.argv
    ks = sorted(ks, key=lambda k: (-k["scratch"], -k["vgpr_spill"], k["name"]))
    names = demangled([k["name"] for k in ks])
    for k, n in zip(ks, names):
        if only_scratch and k["scratch"] == 0:
            continue
        print(f"scratch {k['scratch']:5d} B  vgpr {k['vgpr']:3d} agpr {k['agpr']:3d}  spills v{k['vgpr_spill']:4d} s{k['sgpr_spill']:4d}  {n[:150]}")
