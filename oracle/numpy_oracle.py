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


# ------------------------------------------------------------------------------------------------
# "next" row N4: point-noise -> pose covariance (the consistency experiment)
#   cluster noise covariance : /root/reference/src/simulation/toolss.hpp:315-347   (PointCluster::push, POINT_NOISE)
#   Rcov_raw = sum Ls c_cov Ls^T : /root/reference/src/simulation/BAs_left.hpp:342-473 (left_jacobian_point)
#   Rcov = H^-1 Rcov_raw H^-T  : /root/reference/src/simulation/BAs_left.hpp:1089-1096
#   NEES                       : /root/reference/src/simulation/consistency.cpp:159-170
# ------------------------------------------------------------------------------------------------
def cluster_noise_cov(points, pn):
    """c_cov of PointCluster::push: sum_k Bf p_cov Bf^T with p_cov = pn^2 I (toolss.hpp:321-345)."""
    c = np.zeros((9, 9))
    for v in np.asarray(points, dtype=np.float64).reshape(-1, 3):
        Bf = np.array([[2 * v[0], 0, 0], [v[1], v[0], 0], [v[2], 0, v[0]], [0, 2 * v[1], 0], [0, v[2], v[1]],
                       [0, 0, 2 * v[2]], [1, 0, 0], [0, 1, 0], [0, 0, 1]], dtype=np.float64)
        c += Bf @ (pn * pn * np.eye(3)) @ Bf.T
    return c


def cluster_noise_cov_closed_form(cl, pn):
    """the same matrix from the cluster's own moments (it is linear in P, v, N): [...,10] -> [...,9,9]"""
    xx, xy, xz, yy, yz, zz, x, y, z, N = [cl[..., k] for k in range(10)]
    c = np.zeros(cl.shape[:-1] + (9, 9))
    ent = {(0, 0): 4 * xx, (0, 1): 2 * xy, (0, 2): 2 * xz, (0, 6): 2 * x,
           (1, 1): yy + xx, (1, 2): yz, (1, 3): 2 * xy, (1, 4): xz, (1, 6): y, (1, 7): x,
           (2, 2): zz + xx, (2, 4): xy, (2, 5): 2 * xz, (2, 6): z, (2, 8): x,
           (3, 3): 4 * yy, (3, 4): 2 * yz, (3, 7): 2 * y,
           (4, 4): zz + yy, (4, 5): 2 * yz, (4, 7): z, (4, 8): y,
           (5, 5): 4 * zz, (5, 8): 2 * z, (6, 6): N, (7, 7): N, (8, 8): N}
    for (r, k), v in ent.items():
        c[..., r, k] = v
        c[..., k, r] = v
    return pn * pn * c


def _g1(w):
    g = np.zeros((4, 9))
    g[0, [0, 1, 2, 6]] = [w[0], w[1], w[2], w[3]]
    g[1, [1, 3, 4, 7]] = [w[0], w[1], w[2], w[3]]
    g[2, [2, 4, 5, 8]] = [w[0], w[1], w[2], w[3]]
    g[3, [6, 7, 8]] = [w[0], w[1], w[2]]
    return g


def _g2(w):
    g = np.zeros((6, 3))
    g[:3] = hat(w[:3])
    g[3:] = w[3] * np.eye(3)
    return g


def _feature_frame(C_a):
    NN = C_a[3, 3]
    Cn = C_a / NN
    vbar = Cn[:3, 3]
    lam, U = np.linalg.eigh(Cn[:3, :3] - np.outer(vbar, vbar))
    return NN, Cn, lam, U


def point_cov_left(clusters, ccov, fix, poses, coeffs=None, beg=0, end=None):
    """Literal restatement of left_jacobian_point: Rcov_raw = sum_a sum_j Ls c_cov_j Ls^T with the dense
    (6W x 9) Ls of every (feature, observing pose).  ccov [F,W,9,9].  coeffs (None = the reference's 1)
    weight the features like the gradient they differentiate."""
    F, W = clusters.shape[:2]
    end = F if end is None else end
    n = 6 * W
    T, TC, TCT, C = world_moments(clusters, fix, poses)
    obs = clusters[..., 9] > 0
    Sp = np.zeros((3, 4)); Sp[:, :3] = np.eye(3)
    Fm = np.zeros((4, 4)); Fm[3, 3] = 1
    R = np.zeros((n, n))
    for a in range(beg, end):
        NN, Cn, lam, U = _feature_frame(C[a])
        coe = 1.0 if coeffs is None else coeffs[a]
        l = 0
        Ul = np.zeros((6, 4)); Ul[:3, :3] = hat(-U[:, l]); Ul[3:, 3] = U[:, l]
        SpTul = Sp.T @ U[:, l]
        T_FC = [T[p].T - Fm @ Cn for p in range(W)]
        UlTC = [Ul @ TC[a, p] for p in range(W)]
        g2c = [_g2(TC[a, p] @ T_FC[p] @ SpTul) + UlTC[p] @ T_FC[p] @ Sp.T for p in range(W)]
        for j in range(W):
            if not obs[a, j]:
                continue
            g1_TSu = _g1(T[j].T @ SpTul)
            G = np.zeros((3, 9))
            Gkl = T_FC[j].T @ g1_TSu - T[j] @ _g1(Fm @ Cn @ Sp.T @ U[:, l])
            for k in range(3):
                if k != l:
                    G += 1.0 / (lam[l] - lam[k]) / NN * np.outer(U[:, k], U[:, k]) @ Sp @ Gkl
            Ls = np.zeros((n, 9))
            for p in range(W):
                if not obs[a, p]:
                    continue
                Lp = g2c[p] @ G - 1.0 / NN * UlTC[p] @ Fm @ T[j] @ g1_TSu
                if p == j:
                    Lp = Lp + Ul @ T[p] @ _g1(T_FC[p] @ SpTul)
                Ls[6 * p:6 * p + 6] = 2.0 / NN * Lp
            R += coe * coe * Ls @ ccov[a, j] @ Ls.T
    return R


def point_cov_left_factored(clusters, ccov, fix, poses, coeffs=None):
    """The same matrix through the identity the GPU path uses.  With U12 = [u1 u2]:
         Ls_{a,j} block p = At_p Gm_j + [p == j] D_j ,   At_p = [(2/NN) A_p U12 | -(2/NN^2) w_p]  (6x3),
         Gm_j = [u_k^T Gkl_j / ((lam0 - lam_k) NN), k = 1,2 ; m_j]                               (3x9)
       =>  Rcov_raw = X X^T - Y Y^T + blockdiag(S),   X = At Cq + Y,  Y = Rr Cq^-T,  Q = Cq Cq^T,
       Q = sum_j Gm_j c_cov_j Gm_j^T (3x3), Rr block j = D_j c_cov_j Gm_j^T, S_j = D_j c_cov_j D_j^T;
       X, Y in R^{6W x 3F} -- the columns of At are the Hessian's own factor vectors (Appendix A) rescaled.
       Returns (R, X, Y, S[W,6,6])."""
    F, W = clusters.shape[:2]
    n = 6 * W
    T, TC, TCT, C = world_moments(clusters, fix, poses)
    obs = clusters[..., 9] > 0
    X = np.zeros((n, 3 * F)); Y = np.zeros((n, 3 * F)); S = np.zeros((W, 6, 6))
    for a in range(F):
        NN, Cn, lam, U = _feature_frame(C[a])
        coe = 1.0 if coeffs is None else coeffs[a]
        u0 = U[:, 0]
        vbar = Cn[:3, 3]
        Ul = np.zeros((6, 4)); Ul[:3, :3] = hat(-u0); Ul[3:, 3] = u0
        At = np.zeros((n, 3)); Rr = np.zeros((n, 3)); Q = np.zeros((3, 3))
        for j in range(W):
            if not obs[a, j]:
                continue
            Rj, pj = T[j, :3, :3], T[j, :3, 3]
            Pw, b, N = TCT[a, j, :3, :3], TCT[a, j, :3, 3], TCT[a, j, 3, 3]
            Mtop = Pw - np.outer(b, vbar)
            cvec = b - N * vbar
            m0, s0 = Mtop @ u0, cvec @ u0
            for k in (1, 2):
                uk = U[:, k]
                At[6 * j:6 * j + 3, k - 1] = 2.0 / NN * (np.cross(m0, uk) + np.cross(Mtop @ uk, u0))
                At[6 * j + 3:6 * j + 6, k - 1] = 2.0 / NN * (s0 * uk + (cvec @ uk) * u0)
            At[6 * j:6 * j + 3, 2] = -2.0 / NN / NN * np.cross(b, u0)
            At[6 * j + 3:6 * j + 6, 2] = -2.0 / NN / NN * N * u0
            r3 = Rj.T @ u0
            g1a = _g1(np.append(r3, pj @ u0))
            Gkl3 = Rj @ g1a[:3] + np.outer(pj - vbar, g1a[3])
            Gkl3[:, 6:9] -= (vbar @ u0) * Rj
            Gm = np.vstack([U[:, 1] @ Gkl3 / ((lam[0] - lam[1]) * NN), U[:, 2] @ Gkl3 / ((lam[0] - lam[2]) * NN), g1a[3]])
            g1t = _g1(np.append(r3, (pj - vbar) @ u0))
            D = 2.0 / NN * np.vstack([hat(-u0) @ (Rj @ g1t[:3] + np.outer(pj, g1t[3])), np.outer(u0, g1t[3])])
            Sg = ccov[a, j] @ Gm.T                                                    # 9x3
            Q += Gm @ Sg
            Rr[6 * j:6 * j + 6] = D @ Sg
            S[j] += coe * coe * D @ ccov[a, j] @ D.T
        # Q = Cq Cq^T; a vanished pivot (degenerate feature) drops its column (pseudo-inverse)
        Cq = np.zeros((3, 3)); Ci = np.zeros((3, 3))
        for c in range(3):
            d = Q[c, c] - Cq[c, :c] @ Cq[c, :c]
            if Q[c, c] > 0 and d > 1e-12 * Q[c, c]:
                Cq[c, c] = np.sqrt(d)
                Cq[c + 1:, c] = (Q[c + 1:, c] - Cq[c + 1:, :c] @ Cq[c, :c]) / Cq[c, c]
        live = np.diag(Cq) > 0
        Ci[np.ix_(live, live)] = np.linalg.inv(Cq[np.ix_(live, live)])
        Ya = Rr @ Ci.T
        X[:, 3 * a:3 * a + 3] = coe * (At @ Cq + Ya)
        Y[:, 3 * a:3 * a + 3] = coe * Ya
    R = X @ X.T - Y @ Y.T
    for j in range(W):
        R[6 * j:6 * j + 6, 6 * j:6 * j + 6] += S[j]
    return R, X, Y, S


def pose_cov(H, Rraw):
    """Rcov = H^-1 Rcov_raw H^-T (BAs_left.hpp:1094-1095)"""
    Hi = np.linalg.inv(H)
    return Hi @ Rraw @ Hi.T


def nees(poses_est, poses_gt, Rcov):
    """consistency.cpp:159-170: err_i = [Log(Rgt Rest^T); -Rgt Rest^T p_est + p_gt], NEES = err^T Rcov^-1 err"""
    Re, Rg = pose_R(poses_est), pose_R(poses_gt)
    pe, pg = pose_p(poses_est), pose_p(poses_gt)
    err = np.zeros(6 * Re.shape[0])
    for i in range(Re.shape[0]):
        dR = Rg[i] @ Re[i].T
        c = np.clip((np.trace(dR) - 1) / 2, -1, 1)
        th = np.arccos(c)
        k = np.array([dR[2, 1] - dR[1, 2], dR[0, 2] - dR[2, 0], dR[1, 0] - dR[0, 1]])
        err[6 * i:6 * i + 3] = 0.5 * k if th < 1e-9 else 0.5 * th / np.sin(th) * k
        err[6 * i + 3:6 * i + 6] = -dR @ pe[i] + pg[i]
    return float(err @ np.linalg.solve(Rcov, err)), err
