"""ROS/PCL-free real-world pipeline: what the reference's benchmark_realworld driver
does around the optimizer (src/benchmark/benchmark_realworld.cpp:144-218) -- read alidarPose.csv and the
binary PCD scans, express poses relative to pose 0, associate points to plane features by adaptive
voxelisation (bavoxel.hpp's cut_voxel / recut / tras_opt, on the device: balm_associate, csrc/kernels_voxel.hip;
SURVEY.md 8f N3) -- feeding the GPU path through the C ABI.  Nothing runs on the CPU between the files and the result.

    python -m balm_amd.realworld /path/to/datas/benchmark_realworld [--voxel 2.0]
"""
import ctypes as C
import os
import sys
import time

import numpy as np

from . import scene


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _lib():
    L = scene.host_lib()
    L.balm_read_pcd_xyz.restype = C.c_long
    return L


def read_pose_csv(path, max_poses=100000):
    buf = np.zeros((max_poses, 12))
    stamps = np.zeros(max_poses)
    n = _lib().balm_read_pose_csv(path.encode(), max_poses, _p(buf), _p(stamps))
    if n < 0:
        raise FileNotFoundError(path)
    return buf[:n].copy(), stamps[:n].copy()


def read_pcd_xyz(path):
    L = _lib()
    n = L.balm_read_pcd_xyz(path.encode(), None, C.c_long(0))
    if n < 0:
        raise IOError("cannot read binary PCD %s (%d)" % (path, n))
    xyz = np.zeros((n, 3), dtype=np.float32)
    got = L.balm_read_pcd_xyz(path.encode(), _p(xyz), C.c_long(n))
    return xyz[:got]


def relative_to_first(poses):
    """benchmark_realworld.cpp:163-168: p_i <- R_0^T (p_i - p_0), R_i <- R_0^T R_i"""
    R = poses[:, :9].reshape(-1, 3, 3).transpose(0, 2, 1)
    p = poses[:, 9:]
    Rn = np.einsum("ji,njk->nik", R[0], R)
    pn = (p - p[0]) @ R[0]
    out = np.zeros_like(poses)
    out[:, :9] = Rn.transpose(0, 2, 1).reshape(-1, 9)
    out[:, 9:] = pn
    return out


# the consistency driver's association rules: src/simulation/BAs_left.hpp:18-23 (layer_limit 0, thresholds 1/64,
# min_ps 10), :674 (max point-to-plane distance 1 mm, lambda2/lambda1 < 25, lambda0 < 1e-10), consistency.cpp:125-131
# (first scan marginalised into fix clusters), BAs_left.hpp:38 (no observer minimum)
SIM_RULES = dict(voxel_size=1.0, eigen_thresholds=(1.0 / 64, 1.0 / 64, 1.0 / 64), layer_limit=0, min_ps=10,
                 strict=(0.001, 25.0, 1e-10), fix_frames=1, min_observers=0)


def associate_gpu(ctx, frames_xyz, poses, voxel_size=2.0, eigen_thresholds=(1.0 / 16, 1.0 / 16, 1.0 / 9), layer_limit=2,
                  min_ps=15, strict=None, fix_frames=0, min_observers=2, want_points=False, want_features=True):
    """frames_xyz: list of [n_i,3] float32 body-frame scans; poses [W,12].  Defaults = benchmark_realworld.cpp:183-185 +
    launch/benchmark_realworld.launch:4; `**SIM_RULES` = the consistency driver's rules (W then counts the scans after
    the marginalised ones).  The association runs on the device through balm_associate; the
    features are installed in `ctx` (a context for len(frames_xyz) - fix_frames poses).  -> (F, n_root_voxels,
    feature tuple or None); the point -> feature map refers to the concatenation of the scans."""
    xyz = np.concatenate([np.ascontiguousarray(f, dtype=np.float32).reshape(-1, 3) for f in frames_xyz])
    fid = np.concatenate([np.full(f.shape[0], i, dtype=np.int32) for i, f in enumerate(frames_xyz)])
    return ctx.associate(xyz, fid, poses, voxel_size, eigen_thresholds, min_ps, want_features, layer_limit, min_observers,
                         fix_frames, strict, want_points)


def canonical_order(clusters, coeffs):
    """Order-independent comparison key for feature sets (hash-map order on the host, key order on the device)."""
    first = np.argmax(clusters[:, :, 9] > 0, axis=1)
    head = clusters[np.arange(clusters.shape[0]), first]
    return np.lexsort((head[:, 8], head[:, 7], head[:, 6], first, coeffs))


def load_window(data_dir, max_poses=None):
    poses, stamps = read_pose_csv(os.path.join(data_dir, "alidarPose.csv"))
    if max_poses:
        poses = poses[:max_poses]
    frames = [read_pcd_xyz(os.path.join(data_dir, "full%d.pcd" % m)) for m in range(poses.shape[0])]
    return relative_to_first(poses), frames


SHIPPED_WINDOW_NPZ = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "datasets", "realworld_w177.npz")


def end_to_end(npz_path=SHIPPED_WINDOW_NPZ, device=0, reps=5):
    """benchmark_realworld.cpp:183-218 on the shipped window, from HOST memory to optimised poses, through the C ABI: scans in
    pageable numpy arrays, one per scan -> balm_associate_scans (pinned-ring upload + device association) -> balm_damping_iter with the driver's
    constants.  `npz_path`: xyz [n,3] float32 in scan order, counts [W], poses [W,12] (tools/make_realworld_fixture.py writes it
    from datas/benchmark_realworld with this module's readers), optionally ref_poses / ref_log = the reference's own result.
    First repetition = cold (arena, pinned ring, code objects); the figures are the median of the others.  -> dict."""
    from . import capi
    d = np.load(npz_path)
    xyz = np.ascontiguousarray(d["xyz"], dtype=np.float32).reshape(-1, 3)
    counts = np.asarray(d["counts"]).astype(np.int64)
    poses = np.ascontiguousarray(d["poses"], dtype=np.float64)
    W = int(counts.shape[0])
    offs = np.concatenate([[0], np.cumsum(counts)])
    scans = [xyz[offs[i]:offs[i + 1]] for i in range(W)]          # one container per scan, as the driver holds them (here: packed xyz)
    ctx = capi.Context(W, device, capi.FLAG_TIMING)
    rows = []
    out = lg = None
    F = nroots = 0
    for rep in range(reps + 1):
        ctx.reset_timing()
        t0 = time.perf_counter()
        F, nroots, _ = ctx.associate_scans(scans, poses, 2.0, want_features=False)
        t1 = time.perf_counter()
        out, lg = ctx.damping_iter(poses, form=capi.FORM_LEFT, u0=0.01, max_iter=10, min_planes=20)
        t2 = time.perf_counter()
        tm = ctx.timing()
        rows.append(dict(total=(t2 - t0) * 1e3, associate=(t1 - t0) * 1e3, lm=(t2 - t1) * 1e3, upload=tm["upload"][0],
                         assoc_device=tm["voxel"][0]))
    ctx.close()
    med = {k: float(np.median([r[k] for r in rows[1:]])) for k in rows[0]}
    up_bytes = xyz.nbytes                                          # (no per-point scan index on this route: 177 counts)
    res = {
        "what": "BASELINE configs[4] as shipped (datas/benchmark_realworld: %d scans, %d points): host scans -> balm_associate_scans -> "
                "balm_damping_iter (u0=0.01, <=10 it., >=20 planes/pose) -> poses, wall clock through the C ABI from pageable host "
                "memory; median of %d warm repetitions" % (W, xyz.shape[0], reps),
        "scans": W, "points": int(xyz.shape[0]), "root_voxels": int(nroots), "features": int(F), "lm_iterations": int(len(lg)),
        "ms_total": med["total"], "ms_associate_call": med["associate"], "ms_lm_call": med["lm"],
        "ms_upload": med["upload"], "upload_bytes": int(up_bytes), "upload_gb_per_s": up_bytes / med["upload"] / 1e6 if med["upload"] > 0 else None,
        "ms_associate_device": med["assoc_device"], "ms_cold_first_call": rows[0]["total"],
        "lm_ms_per_iteration": med["lm"] / max(len(lg), 1), "final_residual": float(lg[-1, 1]),
    }
    if "ref_poses" in d.files:
        ref, rl = d["ref_poses"], d["ref_log"]
        rot = tr = 0.0
        for x, y in zip(out, ref):
            D = x[:9].reshape(3, 3) @ y[:9].reshape(3, 3).T
            c = min(1.0, max(-1.0, (np.trace(D) - 1.0) * 0.5))
            sn = 0.5 * np.linalg.norm([D[2, 1] - D[1, 2], D[0, 2] - D[2, 0], D[1, 0] - D[0, 1]])
            rot = max(rot, float(np.arctan2(sn, c)))
            tr = max(tr, float(np.linalg.norm(x[9:] - y[9:])))
        res["vs_reference"] = {"max_rot_rad": rot, "max_trans_m": tr, "iterations_reference": int(len(rl)),
                               "ok": bool(rot <= 1e-5 and tr <= 1e-4 and len(rl) == len(lg)),
                               "what": "final poses against the reference's compiled cut_voxel/recut/tras_opt + BALM2::damping_iter on the same files"}
    return res


CPP_E2E_EXE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "bin", "shim_realworld_e2e")


def write_window_bin(npz_path, out_path):
    """the window file tests/cpp/shim_realworld_e2e.cpp reads: int32 W, int32 has_ref, int64 n | int64 counts[W] | double poses[W*12]
    | double ref_poses[W*12] (if has_ref) | float32 xyz[n*3]"""
    d = np.load(npz_path)
    xyz = np.ascontiguousarray(d["xyz"], dtype=np.float32).reshape(-1, 3)
    counts = np.asarray(d["counts"]).astype(np.int64)
    has_ref = "ref_poses" in d.files
    with open(out_path, "wb") as f:
        f.write(np.array([counts.shape[0], int(has_ref)], dtype=np.int32).tobytes())
        f.write(np.array([xyz.shape[0]], dtype=np.int64).tobytes())
        f.write(counts.tobytes())
        f.write(np.ascontiguousarray(d["poses"], dtype=np.float64).tobytes())
        if has_ref:
            f.write(np.ascontiguousarray(d["ref_poses"], dtype=np.float64).tobytes())
        f.write(xyz.tobytes())
    return out_path


def end_to_end_cpp(npz_path=SHIPPED_WINDOW_NPZ, reps=5, exe=CPP_E2E_EXE, features_out=None, env=None, late=False):
    """The same window through the C++ side of the boundary: tests/cpp/shim_realworld_e2e.cpp = the reference's translation unit
    (tools.hpp, bavoxel.hpp) + include/balm_shim.hpp, scans held as vector<pcl::PointCloud<PointXYZINormal>::Ptr> (48-byte
    elements), `BALM2_HIP::associate(pl_fulls, x_buf); damping_iter(x_buf)` timed on the caller's thread, cold (first use in a fresh
    process, context creation apart) and warm (median).  `late`: the optimizer object is declared after the scans are read (where the
    reference declares `BALM2 opt;`), not first thing in main -- nothing of the device's start-up overlaps.  The binary is built where the reference's headers are
    (tests/cpp/build_shim_driver.sh -> tools/bin/, travels); -> dict, or None when it is not there."""
    import subprocess
    import tempfile
    if not os.path.exists(exe):
        return None
    with tempfile.TemporaryDirectory() as td:
        win = write_window_bin(npz_path, os.path.join(td, "window.bin"))
        cmd = [exe, win, str(int(reps)), features_out or "-"] + (["late"] if late else [])
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("SHIM_E2E ")]
    if not line:
        raise RuntimeError("shim_realworld_e2e failed (rc %d): %s %s" % (out.returncode, out.stdout[-500:], out.stderr[-500:]))
    kv = dict(t.split("=", 1) for t in line[0].split()[1:])
    return {
        "what": "the same window through include/balm_shim.hpp from C++: %s scans as pcl::PointCloud<PointXYZINormal> (%s-byte elements) -> "
                "BALM2_HIP::associate(pl_fulls, x_buf) -> BALM2_HIP::damping_iter(x_buf), steady_clock on the caller's thread; "
                "cold = first use in a fresh process, warm = median of %s further calls" % (kv["scans"], kv["point_bytes"], kv["reps"]),
        "optimizer_object_declared": kv["declared"] + (" in main: the device start-up runs behind the reading of the scans" if kv["declared"] == "first"
                                                         else " (after the scans are read, benchmark_realworld.cpp:217: no overlap)"),
        "clouds_on_numa_nodes": kv.get("clouds_on_nodes"),      # how many of the clouds the scheduler placed on node 0/1/2/3 (profiles/r06_cpp_leg_numa.txt)
        "features": int(kv["features"]), "lm_iterations": int(kv["lm_iterations"]),
        "ms_total": float(kv["total"]), "ms_associate_call": float(kv["associate"]), "ms_lm_call": float(kv["lm"]),
        "ms_upload": float(kv["upload"]), "ms_associate_device": float(kv["assoc_device"]),
        "ms_cold_create": float(kv["cold_create"]), "ms_cold_associate": float(kv["cold_associate"]), "ms_cold_lm": float(kv["cold_lm"]),
        "ms_cold_first_call": float(kv["cold_associate"]) + float(kv["cold_lm"]),
        "vs_reference": {"max_rot_rad": float(kv["max_rot"]), "max_trans_m": float(kv["max_trans"]),
                         "ok": bool(out.returncode == 0 and 0 <= float(kv["max_rot"]) <= 1e-5 and 0 <= float(kv["max_trans"]) <= 1e-4)},
    }


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("data_dir", help="directory with alidarPose.csv + full<k>.pcd; or --npz")
    ap.add_argument("--npz", action="store_true", help="data_dir is a window file as end_to_end() reads it: print its timing dict")
    ap.add_argument("--voxel", type=float, default=2.0)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--gpu-assoc", action="store_true", help="(default since round 2; accepted for old command lines)")
    ap.add_argument("--out", default=None, help="write optimised poses (W x 12) here as .npy")
    a = ap.parse_args(argv)
    from . import capi
    if a.npz:
        import json
        res = end_to_end(a.data_dir, a.device)
        res["cpp_shim"] = end_to_end_cpp(a.data_dir)
        print(json.dumps(res))
        return 0
    t = time.time()
    poses, frames = load_window(a.data_dir)
    t_read = time.time() - t
    W = poses.shape[0]
    ctx = capi.Context(W, a.device)
    t = time.time()
    F = associate_gpu(ctx, frames, poses, a.voxel, want_features=False)[0]
    t_assoc = time.time() - t
    print("The size of poses: %d" % W)                                   # benchmark_realworld.cpp:171
    print("read %d points in %.1f s; %d plane features in %.1f s" % (sum(f.shape[0] for f in frames), t_read, F, t_assoc))
    if F < 3 * W:                                                        # :209-215
        print("Initial error too large.\nPlease loose plane determination criteria for more planes.\n"
              "The optimization is terminated.")
        return 1
    t = time.time()
    out, lg = ctx.damping_iter(poses, form=capi.FORM_LEFT, u0=0.01, max_iter=10, min_planes=20, verbose=True)
    print("optimised in %d LM iterations, %.2f ms" % (len(lg), (time.time() - t) * 1e3))
    if a.out:
        np.save(a.out, out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
