#!/usr/bin/env python3
"""Registers / scratch / LDS of every kernel in liblegged_hip.so (from the code object's metadata notes).
    python tools/kernel_resources.py [filter]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.environ.get("LG_HIP_LIB") or os.path.join(ROOT, "legged_gym_dev_amd", "lib", "liblegged_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    with tempfile.TemporaryDirectory() as d:
        # the fat binary sits in .hip_fatbin: extract with objcopy, then unbundle
        fat = os.path.join(d, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", SO, fat])
        data = open(fat, "rb").read()
        # every translation unit contributes one bundle; split on the bundle magic
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        offs = [m.start() for m in re.finditer(re.escape(magic), data)]
        rows = []
        for i, o in enumerate(offs):
            chunk = data[o:offs[i + 1] if i + 1 < len(offs) else len(data)]
            b = os.path.join(d, f"b{i}.bin")
            open(b, "wb").write(chunk)
            co = os.path.join(d, f"b{i}.co")
            r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + b,
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], capture_output=True)
            if r.returncode or not os.path.isfile(co):
                continue
            txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
            for blk in txt.split("- .agpr_count:")[1:]:
                g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
                name = g("name")
                dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                rows.append((dem, g("vgpr_count"), blk.split()[0], g("sgpr_count"), g("private_segment_fixed_size"),
                             g("group_segment_fixed_size")))
        print(f"{'kernel':100s} vgpr agpr sgpr scratch lds")
        for r in sorted(rows):
            if flt in r[0]:
                print(f"{r[0][:100]:100s} {r[1]:>4s} {r[2]:>4s} {r[3]:>4s} {r[4]:>7s} {r[5]:>6s}")


if __name__ == "__main__":
    main()
