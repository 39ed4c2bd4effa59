#!/usr/bin/env python3
"""ISA lint of the shipped library (CPU only: llvm-objdump on the gfx950 code objects inside libekfslam_hip.so).

The store-data hazard (`stb16` in ekf_kernels.hip): on gfx950 a `buffer_store_dwordx4 ... offen` with an
SGPR soffset reads its data VGPRs in two passes, and a vector instruction issued right behind it that writes one of them
can overtake the second pass.  LLVM's hazard recogniser only covers the forms without an SGPR soffset, so the kernel
follows every such store with an `s_nop 1` that is ordered behind the store (memory clobber) and in front of every
instruction that overwrites the store's data registers (they are inputs of the asm statement).  This tool checks the
machine code that actually ships: walking forward from every such store, no instruction may write a VGPR of the
store's data tuple before at least WAIT_STATES = 2 wait states have gone by (`s_nop N` counts N + 1, any other
instruction 1), and no branch may come before that either.

The DPP hazards (round 6: `fnmac_row_bcast` in ekf_cadence.hip, v_fmac_f64_dpp ... row_newbcast written as inline assembly, which
the compiler's hazard recogniser does not look into): a DPP operation that reads a VGPR through the DPP path needs 2 wait states
behind a VALU instruction that wrote that VGPR, and 5 behind a VALU instruction that wrote EXEC (v_cmpx).  The kernel feeds its
DPP operations from LDS reads, never from VALU results; this tool checks that nothing the compiler put in between (a copy, a
v_cmpx) breaks that in the machine code that ships.

  python3 tools/isa_lint.py [path/to/libekfslam_hip.so]      exit status 1 on a violation
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib_path):
    """The gfx950 code objects of every offload bundle in the shared library (one bundle per translation unit)."""
    data = open(lib_path, "rb").read()
    out, pos = [], 0
    while True:
        i = data.find(MAGIC, pos)
        if i < 0:
            return out
        n = struct.unpack_from("<Q", data, i + len(MAGIC))[0]
        o = i + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, o)
            o += 24
            triple = data[o:o + tl].decode()
            o += tl
            if "gfx950" in triple and size:
                out.append(data[i + off:i + off + size])
        pos = i + len(MAGIC)


def disassemble(lib_path):
    """[(kernel symbol, [instruction text, ...])] over all code objects."""
    funcs = []
    for co in code_objects(lib_path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            text = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], check=True, capture_output=True,
                                  text=True).stdout
        cur = None
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                cur = (m.group(1), [])
                funcs.append(cur)
            elif cur is not None and line.startswith("\t"):
                cur[1].append(line.split("//")[0].strip())
    return funcs


def vgpr_range(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return int(m.group(1)), int(m.group(2))
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return int(m.group(1)), int(m.group(1))
    return None


def writes_vgpr(ins, lo, hi):
    """Whether a vector / LDS / memory-load instruction's destination overlaps v[lo:hi] (first operand = destination)."""
    op, _, rest = ins.partition(" ")
    if not rest or op.startswith(("s_", "buffer_store", "global_store", "ds_write", "scratch_store", "flat_store")):
        return False
    r = vgpr_range(rest.split(",")[0].strip())
    return r is not None and not (r[1] < lo or r[0] > hi)


WAIT_STATES = 2


def wait_states(ins):
    m = re.fullmatch(r"s_nop (\d+)", ins)
    return int(m.group(1)) + 1 if m else 1


def lint(lib_path):
    """Returns (stores checked, stores whose NEXT instruction is not their s_nop, violations as text)."""
    stores = gaps = 0
    bad = []
    for name, ins in disassemble(lib_path):
        for k, text in enumerate(ins):
            if not text.startswith("buffer_store_dwordx4"):
                continue
            ops = [t.strip() for t in text.split(" ", 1)[1].split(",")]
            soffset = ops[3].split()[0] if len(ops) > 3 else "off"
            if not (re.fullmatch(r"s\d+", soffset) and "offen" in text):
                continue                                   # forms the compiler's own hazard recogniser covers
            stores += 1
            lo, hi = vgpr_range(ops[0])
            if k + 1 >= len(ins) or not re.fullmatch(r"s_nop [1-9]\d*", ins[k + 1]):
                gaps += 1
            waited, j = 0, k + 1
            while waited < WAIT_STATES:
                if j >= len(ins):
                    bad.append(f"{name}: `{text}` runs off the end of the function within {WAIT_STATES} wait states")
                    break
                if ins[j].startswith(("s_branch", "s_cbranch", "s_setpc", "s_endpgm")):
                    if not ins[j].startswith("s_endpgm"):
                        bad.append(f"{name}: `{text}` is followed by `{ins[j]}` after {waited} wait state(s)")
                    break
                if writes_vgpr(ins[j], lo, hi):
                    bad.append(f"{name}: `{text}` is followed by `{ins[j]}` after {waited} wait state(s)")
                    break
                waited += wait_states(ins[j])
                j += 1
    return stores, gaps, bad


def is_valu(ins):
    return ins.startswith("v_") and not ins.startswith(("v_readlane", "v_readfirstlane", "v_nop"))


def lint_dpp(lib_path):
    """Returns (DPP operations checked, violations as text): walking BACKWARD from every *_dpp instruction, no VALU instruction
    within 2 wait states may have written its DPP source (the first source operand), none within 5 may have written EXEC; a
    label boundary cannot be seen in this listing, so a branch TARGET in between is not modelled -- branches themselves count as
    one wait state like any other instruction."""
    checked = 0
    bad = []
    for name, ins in disassemble(lib_path):
        for k, text in enumerate(ins):
            op, _, rest = text.partition(" ")
            if not op.endswith("_dpp"):
                continue
            ops = [t.strip() for t in rest.split(",")]
            if len(ops) < 2:
                continue
            src = vgpr_range(ops[1].split()[0].lstrip("-|").rstrip("|"))
            if src is None:
                continue
            checked += 1
            waited, j = 0, k - 1
            while j >= 0 and waited < 5:
                prev = ins[j]
                if is_valu(prev):
                    if waited < 2 and writes_vgpr(prev, src[0], src[1]):
                        bad.append(f"{name}: `{text}` reads what `{prev}` wrote {waited} wait state(s) earlier")
                        break
                    if prev.startswith("v_cmpx"):
                        bad.append(f"{name}: `{text}` comes {waited} wait state(s) behind `{prev}` (EXEC)")
                        break
                waited += wait_states(prev)
                j -= 1
    return checked, bad


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "slam-duckietown_amd", "libekfslam_hip.so")
    n, gaps, bad = lint(path)
    print(f"{path}: {n} buffer_store_dwordx4 (offen + SGPR soffset), {gaps} with other instructions in front of their s_nop "
          f"(harmless: they only add wait states), {len(bad)} violations")
    for b in bad:
        print("  " + b)
    nd, bad_d = lint_dpp(path)
    print(f"{path}: {nd} DPP operations, {len(bad_d)} violations")
    for b in bad_d:
        print("  " + b)
    sys.exit(1 if bad or bad_d else 0)
