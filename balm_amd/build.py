"""In-tree build of the native libraries (no JIT cache: the built .so files travel with the repo).

  balm_amd/lib/libbalm_hip.so    HIP kernels + C ABI (include/balm_hip.h), gfx950 only
  balm_amd/lib/libbalm_scene.so  host-only synthetic scene generator
"""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIBDIR = os.path.join(_HERE, "lib")
HIP_SOURCES = ["kernels_accum.hip", "kernels_solve.hip", "kernels_build.hip", "kernels_voxel.hip", "kernels_cov.hip", "kernels_syrk_i8.hip", "balm_multi.hip", "balm_capi.hip"]
# association decisions must reproduce the reference's un-fused float/double arithmetic bit for bit
EXTRA_FLAGS = {"kernels_voxel.hip": ["-ffp-contract=off"]}
HIP_DEPS = ["balm_internal.h", "host_stage.h", "syrk_mfma_asm.inc", "kernels_window.inc", "kernels_chain.inc", "kernels_small.inc", os.path.join("..", "..", "include", "balm_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-Wall",
             "-Wno-unused-value", "-Wno-unused-result", "-Wno-unused-function"]


def _newer(src, dst):
    return (not os.path.exists(dst)) or os.path.getmtime(src) > os.path.getmtime(dst)


def build_hip(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    so = os.path.join(LIBDIR, "libbalm_hip.so")
    deps = [os.path.join(CSRC, d) for d in HIP_DEPS]
    objs = []
    relink = force or not os.path.exists(so)
    procs = []
    for src in HIP_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer(s, o) or any(_newer(d, o) for d in deps):
            cmd = [HIPCC] + HIP_FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))
            relink = True
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    if relink:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs + ["-ldl", "-pthread"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return so


def build_scene(force=False):
    from . import scene
    return scene.build(force)


def build_all(force=False, verbose=False):
    return build_hip(force, verbose), build_scene(force)


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
