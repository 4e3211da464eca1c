#!/usr/bin/env python3
"""Static audit of the gfx950 code objects in balm_amd/lib/*.so: loops whose body loads from global memory, waits for
vmcnt(0) and only then stores to LDS (or to global memory), i.e. copy loops the compiler left ROLLED -- every iteration is a
full memory round trip.  This is what the helpers of k_ldl_chain did for all of round 3 (~10 serial round trips
per round read as "the memory system is the bound").  Prints kernel, loop position, trip body summary.
   python tools/find_rolled_copies.py [lib.so ...]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(lib, tmp):
    dst = os.path.join(tmp, os.path.basename(lib))
    shutil.copy(lib, dst)
    subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "--offloading", dst], cwd=tmp, stdout=subprocess.DEVNULL)
    return sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if "amdgcn" in f and "gfx950" in f and f.startswith(os.path.basename(lib)))


def audit(obj):
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", obj], capture_output=True, text=True).stdout
    found = []
    kernel, lines = None, []
    def flush():
        if kernel is None:
            return
        # address -> index
        addr = {}
        for i, (a, op, rest) in enumerate(lines):
            addr[a] = i
        for i, (a, op, rest) in enumerate(lines):
            if not op.startswith("s_cbranch"):
                continue
            m = re.search(r"<[^>]*\+0x([0-9a-f]+)>", rest)
            if not m:
                continue
            # objdump prints the target as symbol+offset; instruction addresses are absolute: resolve through the first line
            tgt = base + int(m.group(1), 16)
            j = addr.get(tgt)
            if j is None or j >= i or i - j > 400:
                continue
            body = lines[j:i]
            ops = [o for _, o, _ in body]
            loads = [k for k, o in enumerate(ops) if o.startswith("global_load") and "lds" not in o or o.startswith("buffer_load") or o.startswith("flat_load")]
            stores = [k for k, o in enumerate(ops) if o.startswith("ds_write") or o.startswith("ds_store") or o.startswith("global_store") or o.startswith("flat_store")]
            waits0 = [k for k, (aa, o, r) in enumerate(body) if o == "s_waitcnt" and re.search(r"vmcnt\(0\)", r)]
            if not loads or not stores or not waits0:
                continue
            # a load, then a full wait, then a store that (plausibly) consumes it
            if any(l < w < s for l in loads for w in waits0 for s in stores):
                if any(o.startswith("s_sleep") for o in ops):
                    continue                      # a polling loop, not a copy
                found.append((kernel, a, len(body), len(loads), len(stores), sum(1 for o in ops if o.startswith("v_mfma"))))
    base = 0
    for line in dis.splitlines():
        m = re.match(r"^([0-9a-f]+) <(\S+)>:", line)
        if m:
            flush()
            kernel, lines, base = m.group(2), [], int(m.group(1), 16)
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):(.*)$", line)
        if m and kernel is not None:
            lines.append((int(m.group(3), 16), m.group(1), m.group(2) + " " + m.group(4)))
    flush()
    return found


def demangle(n):
    for tool in (os.path.join(LLVM, "llvm-cxxfilt"), "c++filt"):
        try:
            out = subprocess.run([tool, n], capture_output=True, text=True).stdout.strip()
            if out:
                out = out.replace("(anonymous namespace)::", "")
                return (out[5:] if out.startswith("void ") else out).split("(")[0]
        except Exception:
            pass
    return n


def main():
    libs = sys.argv[1:] or [os.path.join(ROOT, "balm_amd", "lib", f) for f in sorted(os.listdir(os.path.join(ROOT, "balm_amd", "lib"))) if f.endswith(".so")]
    total = 0
    for lib in libs:
        with tempfile.TemporaryDirectory() as tmp:
            for obj in code_objects(lib, tmp):
                for kernel, a, n, nl, ns, nm in audit(obj):
                    if "balm" not in kernel:
                        continue                  # rocPRIM's own kernels
                    total += 1
                    print("%-44s loop ending at 0x%x: %3d instructions, %d loads -> vmcnt(0) -> %d stores per iteration%s" %
                          (demangle(kernel), a, n, nl, ns, ", %d MFMAs" % nm if nm else ""))
    print("%d rolled load -> wait -> store loops" % total)


if __name__ == "__main__":
    main()
