"""numpy/LAPACK twin of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

An independent transcription (vectorised, LAPACK ``eigh``/``solve``) of the same reference math,
used to cross-check oracle/balm_oracle.hpp and to run finite-difference checks:
  left form  : /root/reference/src/benchmark/bavoxel.hpp:304-426 (+ benchmark_virtual.cpp:241-243)
  right form : /root/reference/src/benchmark/bavoxel.hpp:53-158
  residual   : /root/reference/src/benchmark/bavoxel.hpp:428-470
Layouts: clusters [F,W,10] = Pxx Pxy Pxz Pyy Pyz Pzz vx vy vz N ; poses [W,12] = R col-major, p.
"""
import numpy as np


def hat(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], dtype=np.float64)


def exp_so3(w):
    n = np.linalg.norm(w)
    if n < 1e-11:
        return np.eye(3)
    K = hat(np.asarray(w) / n)
    return np.eye(3) + np.sin(n) * K + (1 - np.cos(n)) * K @ K


def pose_R(poses):
    return poses[:, :9].reshape(-1, 3, 3).transpose(0, 2, 1)


def pose_p(poses):
    return poses[:, 9:12]


def make_poses(R, p):
    W = R.shape[0]
    out = np.zeros((W, 12))
    out[:, :9] = R.transpose(0, 2, 1).reshape(W, 9)
    out[:, 9:] = p
    return out


def cluster_mats(cl):
    """[...,10] -> 4x4 homogeneous second-moment matrices [...,4,4]."""
    Co = np.zeros(cl.shape[:-1] + (4, 4))
    idx = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]
    for k, (r, c) in enumerate(idx):
        Co[..., r, c] = cl[..., k]
        Co[..., c, r] = cl[..., k]
    Co[..., :3, 3] = cl[..., 6:9]
    Co[..., 3, :3] = cl[..., 6:9]
    Co[..., 3, 3] = cl[..., 9]
    return Co


def world_moments(clusters, fix, poses):
    W = poses.shape[0]
    T = np.zeros((W, 4, 4))
    T[:, :3, :3] = pose_R(poses)
    T[:, :3, 3] = pose_p(poses)
    T[:, 3, 3] = 1
    Co = cluster_mats(clusters)                      # [F,W,4,4]
    TC = np.einsum("wij,fwjk->fwik", T, Co)
    TCT = np.einsum("fwik,wlk->fwil", TC, T)
    C = TCT.sum(axis=1)
    if fix is not None:
        C = C + cluster_mats(fix)
    return T, TC, TCT, C


def only_residual(clusters, fix, coeffs, poses):
    _, _, _, C = world_moments(clusters, fix, poses)
    NN = C[:, 3, 3]
    Cn = C / NN[:, None, None]
    vbar = Cn[:, :3, 3]
    cov = Cn[:, :3, :3] - vbar[:, :, None] * vbar[:, None, :]
    lam = np.linalg.eigvalsh(cov)
    return float(np.sum(coeffs * lam[:, 0]))


def left_evaluate(clusters, fix, coeffs, poses, return_factors=False):
    F, W = clusters.shape[:2]
    n = 6 * W
    T, TC, TCT, C = world_moments(clusters, fix, poses)
    obs = clusters[..., 9] > 0
    H = np.zeros((n, n))
    g = np.zeros(n)
    res = 0.0
    Gt = np.zeros((n, 3 * F))
    Bd = np.zeros((W, 6, 6))
    for a in range(F):
        coe = coeffs[a]
        NN = C[a, 3, 3]
        Cn = C[a] / NN
        vbar = Cn[:3, 3]
        lam, U = np.linalg.eigh(Cn[:3, :3] - np.outer(vbar, vbar))
        res += coe * lam[0]
        Uk = []
        for k in range(3):
            M = np.zeros((6, 4))
            M[:3, :3] = hat(-U[:, k])
            M[3:, 3] = U[:, k]
            Uk.append(M)
        gk = np.zeros((3, W, 6))
        w = np.zeros((W, 6))
        for i in range(W):
            if not obs[a, i]:
                continue
            tmp = T[i, :3, :].copy()
            tmp[:, 3] -= vbar
            M = TC[a, i] @ tmp.T                       # 4x3
            for k in range(3):
                gk[k, i] = (Uk[k] @ M @ U[:, 0] + Uk[0] @ M @ U[:, k]) / NN
            w[i] = (Uk[0] @ TC[a, i])[:, 3]
            g[6 * i:6 * i + 6] += coe * gk[0, i]
            Ell = hat(M[:3, :3] @ U[:, 0]) @ hat(U[:, 0]) / NN
            B = np.zeros((6, 6))
            B[:3, :3] = Ell + Ell.T
            B += 2.0 / NN * (Uk[0] @ TCT[a, i] @ Uk[0].T)
            Bd[i] += coe * B
        cols = np.stack([np.sqrt(2 * coe) / NN * w.reshape(-1),
                         np.sqrt(2 * coe / (lam[1] - lam[0])) * gk[1].reshape(-1),
                         np.sqrt(2 * coe / (lam[2] - lam[0])) * gk[2].reshape(-1)], axis=1)
        Gt[:, 3 * a:3 * a + 3] = cols
        # literal per-pair accumulation (the form the reference writes)
        wf = w.reshape(-1)
        Hf = -2.0 / NN / NN * np.outer(wf, wf)
        for k in (1, 2):
            gf = gk[k].reshape(-1)
            Hf += 2.0 / (lam[0] - lam[k]) * np.outer(gf, gf)
        H += coe * Hf
    for i in range(W):
        H[6 * i:6 * i + 6, 6 * i:6 * i + 6] += Bd[i]
    if return_factors:
        return H, g, res, Gt, Bd
    return H, g, res


def right_evaluate(clusters, fix, coeffs, poses):
    F, W = clusters.shape[:2]
    n = 6 * W
    R = pose_R(poses)
    p = pose_p(poses)
    _, _, _, C = world_moments(clusters, fix, poses)
    Co = cluster_mats(clusters)
    H = np.zeros((n, n))
    g = np.zeros(n)
    res = 0.0
    I3 = np.eye(3)
    for a in range(F):
        coe = coeffs[a]
        NN = float(int(C[a, 3, 3]))
        vbar = C[a, :3, 3] / C[a, 3, 3]
        lam, U = np.linalg.eigh(C[a, :3, :3] / C[a, 3, 3] - np.outer(vbar, vbar))
        uk = U[:, 0]
        umumT = sum(2.0 / (lam[0] - lam[m]) * np.outer(U[:, m], U[:, m]) for m in (1, 2))
        Auk = np.zeros((W, 3, 6))
        ai = np.zeros((W, 3))
        ni = np.zeros(W)
        seen = []
        for i in range(W):
            if Co[a, i, 3, 3] == 0:
                continue
            seen.append(i)
            Pi = Co[a, i, :3, :3]
            vi = Co[a, i, :3, 3]
            ni[i] = Co[a, i, 3, 3]
            r = R[i].T @ uk
            rh = hat(r)
            vh = hat(vi)
            ai[i] = vh @ r
            ti = p[i] - vbar
            s = uk @ ti
            combo1 = hat(Pi @ r) + vh * s
            combo2 = R[i] @ vi + ni[i] * ti
            Auk[i, :, :3] = ((R[i] @ Pi + np.outer(ti, vi)) @ rh - R[i] @ combo1) / NN
            Auk[i, :, 3:] = (np.outer(combo2, uk) + (combo2 @ uk) * I3) / NN
            jjt = Auk[i].T @ uk
            g[6 * i:6 * i + 6] += coe * jjt
            Hb = Auk[i].T @ umumT @ Auk[i]
            Hb[:3, :3] += 2.0 / NN * (combo1 - rh @ Pi) @ rh - 2.0 / NN / NN * np.outer(ai[i], ai[i]) \
                - 0.5 * hat(jjt[:3])
            HRt = 2.0 / NN * (1.0 - ni[i] / NN) * np.outer(ai[i], uk)
            Hb[:3, 3:] += HRt
            Hb[3:, :3] += HRt.T
            Hb[3:, 3:] += 2.0 / NN * (ni[i] - ni[i] * ni[i] / NN) * np.outer(uk, uk)
            H[6 * i:6 * i + 6, 6 * i:6 * i + 6] += coe * Hb
        for x, i in enumerate(seen):
            for j in seen[x + 1:]:
                bi = np.concatenate([ai[i], ni[i] * uk])
                bj = np.concatenate([ai[j], ni[j] * uk])
                Hb = Auk[i].T @ umumT @ Auk[j] - 2.0 / NN / NN * np.outer(bi, bj)
                H[6 * i:6 * i + 6, 6 * j:6 * j + 6] += coe * Hb
                H[6 * j:6 * j + 6, 6 * i:6 * i + 6] += coe * Hb.T
        res += coe * lam[0]
    return H, g, res


def update_poses(form, poses, dxi):
    R = pose_R(poses)
    p = pose_p(poses)
    W = poses.shape[0]
    Rn = np.zeros_like(R)
    pn = np.zeros_like(p)
    for j in range(W):
        dR = exp_so3(dxi[6 * j:6 * j + 3])
        dt = dxi[6 * j + 3:6 * j + 6]
        if form == 0:
            Rn[j] = dR @ R[j]
            pn[j] = dR @ p[j] + dt
        else:
            Rn[j] = R[j] @ dR
            pn[j] = p[j] + dt
    return make_poses(Rn, pn)


def fd_gradient(form, clusters, fix, coeffs, poses, h=1e-6):
    n = 6 * poses.shape[0]
    g = np.zeros(n)
    for k in range(n):
        d = np.zeros(n)
        d[k] = h
        rp = only_residual(clusters, fix, coeffs, update_poses(form, poses, d))
        rm = only_residual(clusters, fix, coeffs, update_poses(form, poses, -d))
        g[k] = (rp - rm) / (2 * h)
    return g


def fd_hessian(form, clusters, fix, coeffs, poses, h=1e-4):
    """Second-order central differences of the residual along the update map (symmetrised
    second derivative of t -> r(x (+) t d) for d = e_k, e_l, e_k+e_l)."""
    n = 6 * poses.shape[0]

    def r(d):
        return only_residual(clusters, fix, coeffs, update_poses(form, poses, d))

    r0 = r(np.zeros(n))
    dd = np.zeros(n)
    for k in range(n):
        e = np.zeros(n)
        e[k] = h
        dd[k] = (r(e) - 2 * r0 + r(-e)) / h ** 2
    H = np.diag(dd)
    for k in range(n):
        for l in range(k + 1, n):
            e = np.zeros(n)
            e[k] = h
            e[l] = h
            s = (r(e) - 2 * r0 + r(-e)) / h ** 2
            H[k, l] = H[l, k] = 0.5 * (s - dd[k] - dd[l])
    return H
