#!/usr/bin/env python3
"""Build-time check of the invariant k_hessian_syrk rests on (csrc/gen/gen_syrk_asm.py): its 200 accumulator
registers live in physical AGPRs a0..a199 across many separate `asm volatile` statements, and the compiler only
knows about them through clobber lists -- which do not reserve a register between statements.  hipcc is free to
park a spill or a copy in an AGPR inside the k-loop (after a ROCm upgrade, or if the operand ring grows), which
would silently corrupt the Hessian; the cross-compiled, GPU-less build could not notice.

This script disassembles the built object (balm_amd/lib/kernels_accum.o, gfx950 code object) and FAILS unless,
inside k_hessian_syrk and k_hessian_syrk_sparse (the same inlined body):
  * the kernel uses exactly 200 AGPRs, no scratch and no spills (code-object metadata), and
  * every instruction that names an AGPR is one of the generator's three forms:
        v_mfma_f64_16x16x4_f64 a[x:x+7], v.., v.., a[x:x+7]      (accumulate in place)
        v_accvgpr_write_b32 aN, 0                                 (zeroing, 200 of them)
        v_accvgpr_read_b32 vM, aN                                 (read-out after the k-loop)
    and the read-outs all come after the last MFMA (nothing reads or moves an accumulator inside the loop).
Validated with hipcc of ROCm 7.2.0 (clang 20, AMD).  Run by __graft_entry__.build() and tests/test_capi_cpu.py.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
KERNELS = ["k_hessian_syrkEPK", "k_hessian_syrk_sparse"]      # dense and block-sparse entry (same inlined body)


def device_elf(obj, tmp):
    dst = os.path.join(tmp, os.path.basename(obj))
    shutil.copy(obj, dst)
    subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "--offloading", dst], cwd=tmp, stdout=subprocess.DEVNULL)
    for f in os.listdir(tmp):
        if "amdgcn" in f and "gfx950" in f:
            return os.path.join(tmp, f)
    raise RuntimeError("no gfx950 code object in " + obj)


def check_kernel(kernel, notes, dis, verbose):
    problems = []
    # ---- metadata: the kernel's record is the "- .agpr_count ... .name: <mangled>" block containing its name
    recs = re.split(r"\n\s+- \.agpr_count:", notes)
    meta = None
    for r in recs[1:]:
        if re.search(r"\.name:\s+\S*%s\S*" % kernel, r):
            meta = ".agpr_count:" + r
            break
    if meta is None:
        return ["%s not found in the code-object metadata" % kernel]

    def field(k):
        m = re.search(r"\.%s:\s+(\d+)" % k, meta)
        return int(m.group(1)) if m else None
    agprs, scratch, vspill, sspill = field("agpr_count"), field("private_segment_fixed_size"), field("vgpr_spill_count"), field("sgpr_spill_count")
    if agprs != 200:
        problems.append("agpr_count = %s (the generator pins exactly 200: a compiler-chosen AGPR would show up here)" % agprs)
    if scratch or vspill or sspill:
        problems.append("scratch %s B, vgpr spills %s, sgpr spills %s (must all be 0)" % (scratch, vspill, sspill))
    # ---- disassembly of the kernel
    m = re.search(r"\n[0-9a-f]+ <(\S*%s\S*)>:\n(.*?)(?=\n[0-9a-f]+ <|\Z)" % kernel, dis, re.S)
    if not m:
        return problems + ["%s not found in the disassembly" % kernel]
    ins = [l.strip() for l in m.group(2).splitlines() if l.strip()]
    ins = [re.sub(r"\s*//.*$", "", l) for l in ins]
    n_mfma = n_zero = n_read = 0
    last_mfma = first_read = None
    for k, l in enumerate(ins):
        if not re.search(r"\ba\[?\d+", l):
            continue
        mm = re.match(r"v_mfma_f64_16x16x4_f64 a\[(\d+):(\d+)\], v\[\d+:\d+\], v\[\d+:\d+\], a\[(\d+):(\d+)\]$", l)
        if mm:
            lo, hi, lo2, hi2 = map(int, mm.groups())
            if not (lo == lo2 and hi == hi2 and hi == lo + 7 and lo % 8 == 0 and hi < 200):
                problems.append("unexpected MFMA accumulator operands: " + l)
            n_mfma += 1
            last_mfma = k
            continue
        if re.match(r"v_accvgpr_write_b32 a\d+, 0$", l):
            n_zero += 1
            continue
        if re.match(r"v_accvgpr_read_b32 v\d+, a\d+$", l):
            n_read += 1
            first_read = k if first_read is None else first_read
            continue
        problems.append("AGPR touched outside the generated forms: " + l)
    if n_zero != 200:
        problems.append("%d zeroing writes (expected 200)" % n_zero)
    if n_read == 0 or n_read % 200 != 0:
        problems.append("%d accumulator read-outs (expected a multiple of 200)" % n_read)
    if n_mfma == 0 or n_mfma % 25 != 0:
        problems.append("%d MFMAs (expected a multiple of 25)" % n_mfma)
    if first_read is not None and last_mfma is not None and first_read < last_mfma:
        problems.append("an accumulator is read before the last MFMA (instruction %d < %d): the k-loop moves accumulators" % (first_read, last_mfma))
    if verbose:
        print("%s: %d AGPRs, scratch %s, %d MFMAs, %d zeroing writes, %d read-outs -> %s"
              % (kernel, agprs or -1, scratch, n_mfma, n_zero, n_read, "OK" if not problems else "FAILED"))
        for p in problems:
            print("  " + p)
    return ["%s: %s" % (kernel, p) for p in problems]


def check(obj=None, verbose=True):
    obj = obj or os.path.join(ROOT, "balm_amd", "lib", "kernels_accum.o")
    with tempfile.TemporaryDirectory() as tmp:
        elf = device_elf(obj, tmp)
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", elf], capture_output=True, text=True).stdout
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", elf], capture_output=True, text=True).stdout
    problems = []
    for kernel in KERNELS:
        problems += check_kernel(kernel, notes, dis, verbose)
    return problems


if __name__ == "__main__":
    sys.exit(1 if check(sys.argv[1] if len(sys.argv) > 1 else None) else 0)
