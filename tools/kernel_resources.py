#!/usr/bin/env python3
"""VGPR / SGPR / LDS / scratch of the kernels in the shipped library whose (mangled) name contains a pattern:
   python3 tools/kernel_resources.py [pattern] [path/to/libekfslam_hip.so]"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_lint

READOBJ = "/opt/rocm/lib/llvm/bin/llvm-readobj"


def main():
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(here, "slam-duckietown_amd", "libekfslam_hip.so")
    for co in isa_lint.code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            notes = subprocess.run([READOBJ, "--notes", f.name], capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            if pat in name:
                g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)
                print(f"{name[:64]:64s} vgpr {g('vgpr_count'):>3s} sgpr {g('sgpr_count'):>3s} lds {g('group_segment_fixed_size'):>6s} "
                      f"scratch {g('private_segment_fixed_size'):>4s}")


if __name__ == "__main__":
    main()
