"""pose_oracle.py -- numpy restatement of the reference's host-side pose solvers.  TEST INFRASTRUCTURE ONLY: nothing
in the product path imports it (see orb_oracle.h for the rule).  PARITY UNPINNED: the reference holds no fixture for
these functions and cannot be built here (OpenCV / Eigen absent), so this file pins the product against an independent
implementation of the same published algorithms, on LAPACK (numpy.linalg) instead of the product's Jacobi / LDL^T code.

Reference lines followed:
  epnp()               src/PnPsolver.cc:346-876   (compute_pose and everything under it)
  PnpRansac            src/PnPsolver.cc:66-344    (constructor, SetRansacParameters, iterate, Refine, CheckInliers)
  pose_optimization()  src/Optimizer.cc:239-451   on g2o: types_six_dof_expmap.{h:143-215,cpp:266-364}, se3quat.h:223-257,
                       robust_kernel_impl.cpp:78-91, optimization_algorithm_levenberg.cpp:66-170
Rotations are kept as matrices here (the product keeps g2o's quaternions): agreement is to rounding, not bit-exact,
and the tests state their tolerance.
"""
import math
import numpy as np

F32 = np.float32


# ----------------------------------------------------------------------------------------------- EPnP
def _pinv_solve(A, b):
    U, w, Vt = np.linalg.svd(A, full_matrices=False)
    thr = 2 * np.finfo(np.float64).eps * w.sum()
    winv = np.where(w > thr, 1.0 / np.where(w > thr, w, 1.0), 0.0)
    c = U.T @ b
    return Vt.T @ (winv[:, None] * c if c.ndim == 2 else winv * c)


def _absolute_orientation(pcs, pws):
    pc0, pw0 = pcs.mean(0), pws.mean(0)
    abt = (pcs - pc0).T @ (pws - pw0)
    U, w, Vt = np.linalg.svd(abt)
    R = U @ Vt
    if np.linalg.det(R) < 0:
        R[2] = -R[2]
    return R, pc0 - R @ pw0


def epnp(pws, us, fu, fv, uc, vc):
    pws = np.asarray(pws, np.float64).reshape(-1, 3)
    us = np.asarray(us, np.float64).reshape(-1, 2)
    n = len(pws)
    cws = np.zeros((4, 3))
    cws[0] = pws.mean(0)
    d = pws - cws[0]
    ev, evec = np.linalg.eigh(d.T @ d)
    for i in range(3):                                   # descending eigenvalues; axis sign: largest component positive
        ax = evec[:, 2 - i]
        if ax[np.argmax(np.abs(ax))] < 0:
            ax = -ax
        cws[i + 1] = cws[0] + math.sqrt(max(ev[2 - i], 0.0) / n) * ax
    cc = (cws[1:] - cws[0]).T
    ci = _pinv_solve(cc, np.eye(3))
    al = np.zeros((n, 4))
    al[:, 1:] = d @ ci.T
    al[:, 0] = 1.0 - al[:, 1] - al[:, 2] - al[:, 3]
    M = np.zeros((2 * n, 12))
    for j in range(4):
        M[0::2, 3 * j] = al[:, j] * fu
        M[0::2, 3 * j + 2] = al[:, j] * (uc - us[:, 0])
        M[1::2, 3 * j + 1] = al[:, j] * fv
        M[1::2, 3 * j + 2] = al[:, j] * (vc - us[:, 1])
    ev, evec = np.linalg.eigh(M.T @ M)                   # ascending: column 0 = smallest
    v = [evec[:, i].reshape(4, 3) for i in range(4)]     # v[0] <-> ut row 11
    pairs = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]
    dv = np.array([[v[i][a] - v[i][b] for (a, b) in pairs] for i in range(4)])     # 4 x 6 x 3
    L = np.zeros((6, 10))
    for i in range(6):
        g = lambda a, b: float(dv[a][i] @ dv[b][i])
        L[i] = [g(0, 0), 2 * g(0, 1), g(1, 1), 2 * g(0, 2), 2 * g(1, 2), g(2, 2), 2 * g(0, 3), 2 * g(1, 3), 2 * g(2, 3), g(3, 3)]
    rho = np.array([((cws[a] - cws[b]) ** 2).sum() for (a, b) in pairs])

    def approx1():
        b4 = _pinv_solve(L[:, [0, 1, 3, 6]], rho)
        s = -1.0 if b4[0] < 0 else 1.0
        b0 = math.sqrt(s * b4[0])
        return np.array([b0, s * b4[1] / b0, s * b4[2] / b0, s * b4[3] / b0])

    def approx2():
        b3 = _pinv_solve(L[:, [0, 1, 2]], rho)
        if b3[0] < 0:
            be = [math.sqrt(-b3[0]), math.sqrt(-b3[2]) if b3[2] < 0 else 0.0]
        else:
            be = [math.sqrt(b3[0]), math.sqrt(b3[2]) if b3[2] > 0 else 0.0]
        if b3[1] < 0:
            be[0] = -be[0]
        return np.array(be + [0.0, 0.0])

    def approx3():
        b5 = _pinv_solve(L[:, [0, 1, 2, 3, 4]], rho)
        if b5[0] < 0:
            be = [math.sqrt(-b5[0]), math.sqrt(-b5[2]) if b5[2] < 0 else 0.0]
        else:
            be = [math.sqrt(b5[0]), math.sqrt(b5[2]) if b5[2] > 0 else 0.0]
        if b5[1] < 0:
            be[0] = -be[0]
        return np.array(be + [b5[3] / be[0], 0.0])

    quad = [(0, 0), (0, 1), (1, 1), (0, 2), (1, 2), (2, 2), (0, 3), (1, 3), (2, 3), (3, 3)]

    def gauss_newton(be):
        be = be.copy()
        for _ in range(5):
            A = np.zeros((6, 4))
            f = np.zeros(6)
            for k, (a, b) in enumerate(quad):
                f += L[:, k] * be[a] * be[b]
                A[:, a] += L[:, k] * be[b]
                A[:, b] += L[:, k] * be[a]
            if (np.abs(A).max(0) == 0).any():
                continue
            be += np.linalg.lstsq(A, rho - f, rcond=None)[0]
        return be

    def pose(be):
        ccs = sum(be[i] * v[i] for i in range(4))
        pcs = al @ ccs
        if pcs[0, 2] < 0:
            pcs = -pcs
        R, t = _absolute_orientation(pcs, pws)
        pc = pws @ R.T + t
        ue = uc + fu * pc[:, 0] / pc[:, 2]
        ve = vc + fv * pc[:, 1] / pc[:, 2]
        return float(np.sqrt((us[:, 0] - ue) ** 2 + (us[:, 1] - ve) ** 2).mean()), R, t

    sols = [pose(gauss_newton(f())) for f in (approx1, approx2, approx3)]
    N = 0
    if sols[1][0] < sols[0][0]:
        N = 1
    if sols[2][0] < sols[N][0]:
        N = 2
    return sols[N][1], sols[N][2], sols[N][0]


# ----------------------------------------------------------------------------------------------- RANSAC
class PnpRansac:
    def __init__(self, p2d, sigma2, p3d, fx, fy, cx, cy, rand, rand_max):
        self.p2d = np.asarray(p2d, F32).reshape(-1, 2)
        self.sigma2 = np.asarray(sigma2, F32)
        self.p3d = np.asarray(p3d, F32).reshape(-1, 3)
        self.K = (float(F32(fx)), float(F32(fy)), float(F32(cx)), float(F32(cy)))
        self.N = len(self.p2d)
        self.rand, self.rand_max = rand, rand_max
        self.n_iter = self.n_best = 0
        self.best = None
        self.set_parameters()

    def set_parameters(self, prob=0.99, min_inliers=8, max_its=300, min_set=4, eps=0.4, th2=5.991):
        N = self.N
        eps = F32(eps)
        nmin = int(F32(N) * eps)
        nmin = max(nmin, min_inliers, min_set)
        self.min_inliers = nmin
        if eps < F32(nmin) / F32(N):
            eps = F32(nmin) / F32(N)
        self.eps = eps
        if nmin == N:
            nit = 1
        else:
            with np.errstate(all="ignore"):
                v = np.ceil(np.log(np.float64(1 - prob)) / np.log(np.float64(1) - np.float64(eps) ** 3))
            nit = int(v) if -2147483648.0 <= v <= 2147483647.0 else -2 ** 31     # x86 double->int of NaN / out of range
        self.max_its = max(1, min(nit, max_its))
        self.min_set = min_set
        self.max_error = self.sigma2 * F32(th2)

    def _check(self, R, t):
        P = self.p3d.astype(np.float64)
        fx, fy, cx, cy = self.K
        Xc = (P @ R[0] + t[0]).astype(F32)
        Yc = (P @ R[1] + t[1]).astype(F32)
        iz = (1.0 / (P @ R[2] + t[2])).astype(F32)
        ue = cx + fx * Xc.astype(np.float64) * iz
        ve = cy + fy * Yc.astype(np.float64) * iz
        dx = (self.p2d[:, 0] - ue).astype(F32)
        dy = (self.p2d[:, 1] - ve).astype(F32)
        e2 = dx * dx + dy * dy
        return e2 < self.max_error

    @staticmethod
    def _T(R, t):
        T = np.eye(4, dtype=F32)
        T[:3, :3] = R
        T[:3, 3] = t
        return T

    def _pose(self, idx):
        fx, fy, cx, cy = self.K
        R, t, _ = epnp(self.p3d[idx].astype(np.float64), self.p2d[idx].astype(np.float64), fx, fy, cx, cy)
        return R, t

    def iterate(self, n_iterations):
        """-> (Tcw or None, no_more, inliers, n_inliers)"""
        if self.N < self.min_inliers:
            return None, True, None, 0
        cur = 0
        while self.n_iter < self.max_its or cur < n_iterations:
            cur += 1
            self.n_iter += 1
            avail = list(range(self.N))
            pick = []
            for _ in range(self.min_set):
                d = len(avail)
                r = int((self.rand() / (self.rand_max + 1.0)) * d)
                pick.append(avail[r])
                avail[r] = avail[-1]
                avail.pop()
            R, t = self._pose(pick)
            inl = self._check(R, t)
            if inl.sum() >= self.min_inliers:
                if inl.sum() > self.n_best:
                    self.n_best, self.best, self.best_T = int(inl.sum()), inl.copy(), self._T(R, t)
                R, t = self._pose(np.flatnonzero(self.best))
                inl = self._check(R, t)
                if inl.sum() > self.min_inliers:
                    return self._T(R, t), False, inl, int(inl.sum())
        if self.n_iter >= self.max_its and self.n_best >= self.min_inliers:
            return self.best_T, True, self.best, self.n_best
        return None, self.n_iter >= self.max_its, None, 0


# ----------------------------------------------------------------------------------------------- PoseOptimization
def _skew(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])


def _se3_exp(u):
    om, up = u[:3], u[3:]
    th = float(np.linalg.norm(om))
    O = _skew(om)
    if th < 0.00001:
        R = np.eye(3) + O + O @ O
        V = R
    else:
        O2 = O @ O
        R = np.eye(3) + math.sin(th) / th * O + (1 - math.cos(th)) / (th * th) * O2
        V = np.eye(3) + (1 - math.cos(th)) / (th * th) * O + (th - math.sin(th)) / th ** 3 * O2
    return R, V @ up


def _orthonormalise(R):
    U, _, Vt = np.linalg.svd(R)
    return U @ Vt


def pose_optimization(obs, u_right, inv_sigma2, xw, fx, fy, cx, cy, bf, Tcw):
    """-> (Tcw float32 4x4, outlier uint8[n], n_inliers)"""
    obs = np.asarray(obs, F32).reshape(-1, 2).astype(np.float64)
    n = len(obs)
    ur = np.full(n, -1.0) if u_right is None else np.asarray(u_right, F32).astype(np.float64)
    info = np.asarray(inv_sigma2, F32).astype(np.float64)
    xw = np.asarray(xw, F32).reshape(-1, 3).astype(np.float64)
    fx, fy, cx, cy, bf = (float(F32(a)) for a in (fx, fy, cx, cy, bf))
    T0 = np.asarray(Tcw, F32).reshape(4, 4).astype(np.float64)
    R0, t0 = _orthonormalise(T0[:3, :3]), T0[:3, 3].copy()      # the quaternion round trip normalises the rotation
    stereo = ur >= 0
    dm, ds = float(F32(math.sqrt(5.991))), float(F32(math.sqrt(7.815)))
    delta = np.where(stereo, ds, dm)
    outlier = np.zeros(n, np.uint8)
    if n < 3:
        return np.asarray(Tcw, F32).reshape(4, 4).copy(), outlier, 0
    err = np.zeros((n, 3))
    level = np.zeros(n, int)
    robust = np.ones(n, bool)

    def errors(R, t, idx):
        p = xw[idx] @ R.T + t
        e = np.zeros((len(idx), 3))
        m = ~stereo[idx]
        e[m, 0] = obs[idx][m, 0] - (p[m, 0] / p[m, 2] * fx + cx)
        e[m, 1] = obs[idx][m, 1] - (p[m, 1] / p[m, 2] * fy + cy)
        s = ~m
        if s.any():
            iz = (F32(1.0) / p[s, 2].astype(F32)).astype(np.float64)
            u = p[s, 0] * iz * fx + cx
            e[s, 0] = obs[idx][s, 0] - u
            e[s, 1] = obs[idx][s, 1] - (p[s, 1] * iz * fy + cy)
            e[s, 2] = ur[idx][s] - (u - bf * iz)
        return e

    def chi2(idx):
        return (err[idx] ** 2).sum(1) * info[idx]

    def rho(idx):
        c = chi2(idx)
        d = delta[idx]
        out = c > d * d
        s = np.sqrt(np.where(out, c, 1.0))
        r0 = np.where(out, 2 * s * d - d * d, c)
        r1 = np.where(out, d / s, 1.0)
        rb = robust[idx]
        return np.where(rb, r0, c), np.where(rb, r1, 1.0)

    R, t = R0, t0
    n_bad = 0
    for it in range(4):
        R, t = R0.copy(), t0.copy()
        act = np.flatnonzero(level == 0)
        if len(act):
            lam, ni, bad = 0.0, 2.0, 0
            for k in range(10):
                err[act] = errors(R, t, act)
                r0, r1 = rho(act)
                current = ini = float(r0.sum())
                p = xw[act] @ R.T + t
                x, y, iz = p[:, 0], p[:, 1], 1.0 / p[:, 2]
                iz2 = iz * iz
                J = np.zeros((len(act), 3, 6))
                J[:, 0] = np.stack([x * y * iz2 * fx, -(1 + x * x * iz2) * fx, y * iz * fx, -iz * fx, 0 * x, x * iz2 * fx], 1)
                J[:, 1] = np.stack([(1 + y * y * iz2) * fy, -x * y * iz2 * fy, -x * iz * fy, 0 * x, -iz * fy, y * iz2 * fy], 1)
                st = stereo[act]
                J[st, 2] = J[st, 0]
                J[st, 2, 0] -= bf * y[st] * iz2[st]
                J[st, 2, 1] += bf * x[st] * iz2[st]
                J[st, 2, 4] = 0
                J[st, 2, 5] -= bf * iz2[st]
                w = r1 * info[act]
                H = np.einsum("n,ndr,ndc->rc", w, J, J)
                b = -np.einsum("n,ndr,nd->r", w, J, err[act])
                if k == 0:
                    lam, ni, bad = 1e-5 * float(np.abs(np.diag(H)).max()), 2.0, 0
                q, rr = 0, 0.0
                while True:
                    Rb, tb = R, t
                    try:
                        dx = np.linalg.solve(H + lam * np.eye(6), b)
                        ok = bool(np.isfinite(dx).all())
                    except np.linalg.LinAlgError:
                        dx, ok = np.zeros(6), False
                    dR, dt = _se3_exp(dx)
                    R, t = dR @ R, dR @ t + dt
                    err[act] = errors(R, t, act)
                    temp = float(rho(act)[0].sum()) if ok else float("inf")
                    rr = (current - temp) / (float(dx @ (lam * dx + b)) + 1e-3)
                    if rr > 0 and math.isfinite(temp):
                        alpha = min(1.0 - (2 * rr - 1) ** 3, 2.0 / 3.0)
                        lam *= max(1.0 / 3.0, alpha)
                        ni = 2.0
                        current = temp
                    else:
                        lam *= ni
                        ni *= 2
                        R, t = Rb, tb
                    q += 1
                    if not (rr < 0 and q < 10):
                        break
                if q == 10 or rr == 0:
                    break
                bad = bad + 1 if (ini - current) * 1e3 < ini else 0
                if bad >= 3:
                    break
        n_bad = 0
        o = np.flatnonzero(outlier)
        if len(o):
            err[o] = errors(R, t, o)
        c = chi2(np.arange(n)).astype(F32)
        th = np.where(stereo, F32(7.815), F32(5.991)).astype(F32)
        outlier = (c > th).astype(np.uint8)
        level = outlier.astype(int)
        n_bad = int(outlier.sum())
        if it == 2:
            robust[:] = False
        if n < 10:
            break
    T = np.eye(4, dtype=F32)
    T[:3, :3] = R
    T[:3, 3] = t
    return T, outlier, n - n_bad
