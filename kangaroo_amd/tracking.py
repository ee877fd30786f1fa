"""Host side of the tracking loop: the solve / pose-update steps the reference application performs
around roo::PoseRefinementProjectiveIcpPointPlane (applications/kinectfusion/main.cpp:299-343).

The reference does this part with two third-party libraries that are not in its tree (found by CMake,
no pinned version): Eigen (`FullPivLU<Matrix<double,N,N>>::solve`) and Sophus (`SE3d::exp`,
`SO3d::exp`, `inverse`).  Their published algorithms are restated here in float64 numpy:

* LU decomposition with complete pivoting, rank decided by Eigen's default threshold
  (machine epsilon x matrix size x largest pivot), free variables of a rank-deficient system set to 0;
* the SE(3) exponential in closed form: R = I + sin(t)/t W + (1-cos t)/t^2 W^2,
  p = (I + (1-cos t)/t^2 W + (t-sin t)/t^3 W^2) upsilon, tangent vector ordered (upsilon, omega).

Everything here is a handful of 6x6 operations per ICP iteration; the per-pixel work is the HIP kernel
behind `ops.PoseRefinementProjectiveIcpPointPlane`.
"""
import numpy as np

# main.cpp:52 -- iterations per pyramid level (level 0 = full resolution), coarse to fine
DEFAULT_ITS = (1, 0, 2, 3)
MOTION_SIGMA, DEPTH_SIGMA = 0.2, 0.1  # main.cpp:316-318: weak prior on the pose


def full_piv_lu_solve(A, b):
    """x = FullPivLU(A).solve(b) for a small square system (float64)."""
    A = np.array(A, np.float64)
    b = np.array(b, np.float64).reshape(-1)
    n = A.shape[0]
    assert A.shape == (n, n) and b.shape == (n,)
    lu = A.copy()
    rows, cols = np.arange(n), np.arange(n)
    nonzero_pivots, maxpivot = n, 0.0
    for k in range(n):
        sub = np.abs(lu[k:, k:])
        r, c = np.unravel_index(np.argmax(sub), sub.shape)
        biggest = sub[r, c]
        if biggest == 0.0:
            nonzero_pivots = k
            break
        maxpivot = max(maxpivot, biggest)
        r, c = r + k, c + k
        if r != k:
            lu[[k, r], :] = lu[[r, k], :]
            rows[[k, r]] = rows[[r, k]]
        if c != k:
            lu[:, [k, c]] = lu[:, [c, k]]
            cols[[k, c]] = cols[[c, k]]
        if k < n - 1:
            lu[k + 1:, k] /= lu[k, k]
            lu[k + 1:, k + 1:] -= np.outer(lu[k + 1:, k], lu[k, k + 1:])
    thresh = np.finfo(np.float64).eps * n * maxpivot
    rank = int(sum(abs(lu[i, i]) > thresh for i in range(nonzero_pivots)))
    if rank == 0:
        return np.zeros(n)
    c = b[rows].copy()
    for i in range(n):                       # forward substitution with unit-lower L
        c[i] -= lu[i, :i] @ c[:i]
    y = np.zeros(n)
    for i in range(rank - 1, -1, -1):        # back substitution on the leading rank x rank block of U
        y[i] = (c[i] - lu[i, i + 1:rank] @ y[i + 1:rank]) / lu[i, i]
    x = np.zeros(n)
    x[cols] = y
    return x


def hat(w):
    return np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])


def so3_exp(omega):
    omega = np.asarray(omega, np.float64)
    t2 = float(omega @ omega)
    t = np.sqrt(t2)
    W = hat(omega)
    if t < 1e-10:
        return np.eye(3) + W + 0.5 * (W @ W)
    return np.eye(3) + (np.sin(t) / t) * W + ((1.0 - np.cos(t)) / t2) * (W @ W)


def se3_exp(x):
    """4x4 matrix of exp((upsilon, omega))."""
    x = np.asarray(x, np.float64)
    ups, omega = x[:3], x[3:]
    t2 = float(omega @ omega)
    t = np.sqrt(t2)
    W = hat(omega)
    if t < 1e-10:
        V = np.eye(3) + 0.5 * W + (W @ W) / 6.0
    else:
        V = np.eye(3) + ((1.0 - np.cos(t)) / t2) * W + ((t - np.sin(t)) / (t2 * t)) * (W @ W)
    T = np.eye(4)
    T[:3, :3] = so3_exp(omega)
    T[:3, 3] = V @ ups
    return T


def se3_inv(T):
    R, p = T[:3, :3], T[:3, 3]
    Ti = np.eye(4)
    Ti[:3, :3] = R.T
    Ti[:3, 3] = -R.T @ p
    return Ti


def k_matrix(K):
    """ImageIntrinsics::Matrix(): [[fu,0,u0],[0,fv,v0],[0,0,1]]."""
    fu, fv, u0, v0 = [float(v) for v in K]
    return np.array([[fu, 0.0, u0], [0.0, fv, v0], [0.0, 0.0, 1.0]])


def update_pose(T_lp, lss, rotation_only):
    """One Gauss-Newton step on T_lp from a summed system (main.cpp:312-333).  Returns (T_lp, rmse)."""
    JTJ = np.array(lss.JTJ, np.float64) + (DEPTH_SIGMA / MOTION_SIGMA) * np.eye(6)
    JTy = np.array(lss.JTy, np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        rmse = float(np.sqrt(np.float32(lss.sqErr) / np.float32(lss.obs)))   # sqrt(lss.sqErr / lss.obs), float
    if rotation_only:
        x = -1.0 * full_piv_lu_solve(JTJ[3:, 3:], JTy[3:])
        dT = np.eye(4)
        dT[:3, :3] = so3_exp(x)
        T_lp = T_lp @ dT
    else:
        x = -1.0 * full_piv_lu_solve(JTJ, JTy)
        if np.isfinite(x).all():
            T_lp = T_lp @ se3_exp(x)
    return T_lp, rmse


def refine_pose(ops, kin_v, ray_v, ray_n, K_levels, workspace, debug=None, its=DEFAULT_ITS, icp_c=0.1, max_rmse=0.10,
                on_iteration=None):
    """The coarse-to-fine loop of main.cpp:301-336.  kin_v / ray_v / ray_n: per-level float4 images (live vertex
    map, model vertex map and normals from the raycast).  Returns (T_lp 4x4 float64, rmse, tracking_good);
    the caller applies T_wl = T_wl * T_lp^-1 when tracking_good (main.cpp:338-340)."""
    levels = len(K_levels)
    T_lp = np.eye(4)
    rmse, tracking_good = 0.0, True
    for l in range(levels - 1, -1, -1):
        Kd = k_matrix(K_levels[l])
        for _ in range(its[l]):
            KT_lp = (Kd @ T_lp[:3, :]).astype(np.float32)
            T_pl = se3_inv(T_lp)[:3, :].astype(np.float32)
            dbg = None if debug is None else debug.SubImage(0, 0, kin_v[l].w, kin_v[l].h)
            lss = ops.PoseRefinementProjectiveIcpPointPlane(kin_v[l], ray_v[l], ray_n[l], KT_lp, T_pl, icp_c, workspace, dbg)
            T_lp, rmse = update_pose(T_lp, lss, rotation_only=(l == levels - 1 and levels > 1))
            tracking_good = rmse < max_rmse
            if on_iteration is not None:
                on_iteration(l, lss, T_lp, rmse)
    return T_lp, rmse, tracking_good
