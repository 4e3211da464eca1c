#!/usr/bin/env python3
"""Build-time check of what the persistent factorisation kernels' hand-over protocol rests on (kernels_solve.hip: st_wt /
publish / flag_peek; kernels_chain.inc): workgroups on different XCDs exchange tiles and flags through memory, and the L2
caches of the XCDs are not coherent with one another for ordinary accesses.  The kernels therefore rely on the compiler
turning their agent-scope atomics into
    global_store_dwordx2 ... sc1     write-through stores of everything another workgroup will read (tiles, Minv, x blocks)
    global_load_dword(x2) ... sc1    flag / payload polls that do not hit a stale line
    buffer_inv sc1                   the acquire after a flag has been seen
and on NOT getting a buffer_wbl2 (an L2 write-back per publish: 3-5 us each, measured in round 2) anywhere in them.
A ROCm upgrade that changed this lowering would not fail a single CPU test and would show up on the GPU as rare wrong
solutions; this script disassembles the built object (balm_amd/lib/kernels_solve.o, gfx950) and FAILS unless every kernel
below shows the instructions it needs.  Run by __graft_entry__.build() and tests/test_capi_cpu.py.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
# kernel -> minimum counts of (write-through stores, coherent loads, acquire invalidates)
NEEDS = {"k_ldl_chain": (20, 6, 4), "k_ldl_fused": (8, 2, 2), "k_ldl_backsolve": (1, 1, 0)}


def device_elf(obj, tmp):
    dst = os.path.join(tmp, os.path.basename(obj))
    shutil.copy(obj, dst)
    subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "--offloading", dst], cwd=tmp, stdout=subprocess.DEVNULL)
    for f in os.listdir(tmp):
        if "amdgcn" in f and "gfx950" in f:
            return os.path.join(tmp, f)
    raise RuntimeError("no gfx950 code object in " + obj)


def check(obj=None, verbose=True):
    obj = obj or os.path.join(ROOT, "balm_amd", "lib", "kernels_solve.o")
    problems = []
    with tempfile.TemporaryDirectory() as tmp:
        elf = device_elf(obj, tmp)
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", elf], capture_output=True, text=True, check=True).stdout
    # split into functions
    funcs = {}
    cur = None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1)
            funcs[cur] = []
        elif cur is not None:
            funcs[cur].append(line)
    for kern, (need_st, need_ld, need_inv) in NEEDS.items():
        body = None
        for name, lines in funcs.items():
            if re.search(r"\d+%s[A-Z]" % kern, name) and not name.endswith(".kd"):
                body = lines
                break
        if body is None:
            problems.append("%s: not found in the code object" % kern)
            continue
        st = sum(1 for l in body if re.search(r"\bglobal_store_dwordx?2?\b.*\bsc1\b", l))
        ld = sum(1 for l in body if re.search(r"\bglobal_load_dword(x2)?\b.*\bsc1\b", l))
        inv = sum(1 for l in body if re.search(r"\bbuffer_inv\b.*\bsc1\b", l))
        wb = sum(1 for l in body if "buffer_wbl2" in l)
        if verbose:
            print("%-16s write-through stores %3d (>= %d)  coherent loads %3d (>= %d)  buffer_inv sc1 %3d (>= %d)  buffer_wbl2 %d (== 0)"
                  % (kern, st, need_st, ld, need_ld, inv, need_inv, wb))
        if st < need_st:
            problems.append("%s: %d write-through (sc1) stores, expected >= %d" % (kern, st, need_st))
        if ld < need_ld:
            problems.append("%s: %d coherent (sc1) loads, expected >= %d" % (kern, ld, need_ld))
        if inv < need_inv:
            problems.append("%s: %d buffer_inv sc1, expected >= %d" % (kern, inv, need_inv))
        if wb:
            problems.append("%s: %d buffer_wbl2 (an L2 write-back per publish: the protocol avoids them)" % (kern, wb))
        if kern == "k_ldl_chain":
            # the helpers' operand tiles go global -> LDS directly (ch_stage_tile: 2 pieces x 4 tiles in each of the two helper
            # roles).  Written as a plain copy loop the staging stayed ROLLED in the ISA for all of round 3 -- load, vmcnt(0),
            # ds_write per iteration, ~10 serial memory round trips per round; tools/find_rolled_copies.py.
            glds = sum(1 for l in body if "global_load_lds_dwordx4" in l)
            if verbose:
                print("%-16s global_load_lds_dwordx4 %3d (>= 16)" % (kern, glds))
            if glds < 16:
                problems.append("%s: %d global_load_lds_dwordx4, expected >= 16 (the helpers' operand staging)" % (kern, glds))
    return problems


if __name__ == "__main__":
    p = check()
    for x in p:
        print("FAIL:", x)
    sys.exit(1 if p else 0)
