"""Independent numpy restatement of single estimation steps, used only to cross-check the C
oracle's algebra.  Unlike the oracle (and the reference) it never
mutates the clouds: normalisation is applied on the fly, which is also how the HIP path
works, so agreement here validates that design too."""
import numpy as np


def mad_scale(v):
    v = np.abs(np.asarray(v, dtype=np.float64))
    n = v.size
    if n <= 1:
        return 0.0
    return 1.4826 * (1 + 5.0 / (n - 1)) * np.sort(v)[(n - 1) // 2 + 1]


def weight_BT(u):
    return np.where(u <= 16.0, (1 - u * 0.0625) ** 2, 0.0)


def residual2(A, t, c, p, q, P):
    e = (p - c) @ A.T + t - q
    r2 = np.einsum("ni,nij,nj->n", e, P.reshape(-1, 3, 3), e)
    return np.maximum(r2, 0.0)


def jac(dof, x, c):
    n = x.shape[0]
    l = x - c
    J = np.zeros((n, 3, dof))
    if dof == 6:
        J[:, 1, 0], J[:, 2, 0] = -l[:, 2], l[:, 1]
        J[:, 0, 1], J[:, 2, 1] = l[:, 2], -l[:, 0]
        J[:, 0, 2], J[:, 1, 2] = -l[:, 1], l[:, 0]
        J[:, 0, 3] = J[:, 1, 4] = J[:, 2, 5] = 1.0
    else:
        for d in range(3):
            J[:, d, 3 * d:3 * d + 3] = l
            J[:, d, 9 + d] = 1.0
    return J


def normal_equations(dof, from_loc, c, to_loc, to_proj, w, rigid_increment=False, skip=None):
    J = jac(dof, from_loc, c)
    P = to_proj.reshape(-1, 3, 3)
    b = to_loc - from_loc if rigid_increment else to_loc
    if skip is not None:
        w = np.where(w < skip, 0.0, w)
    AtA = np.einsum("nia,nij,njb,n->ab", J, P, J, w)
    Atb = np.einsum("nia,nij,nj,n->a", J, P, b, w)
    return AtA, Atb


def normalize_trans(A, t, c, rm, rf, cm, cf):
    A2 = rm / rf * A
    t2 = t / rf + A2 @ (cm - c) / rm - cf / rf
    return A2, t2


def centres_radii(from_loc, to_loc_all, to_idx):
    cm = from_loc.mean(0)
    rm = np.linalg.norm(from_loc - cm, axis=1).mean()
    uq = np.unique(to_idx)
    cf = to_loc_all[uq].mean(0)
    rf = np.linalg.norm(to_loc_all[uq] - cf, axis=1).mean()
    return cm, rm, cf, rf


def strength_gain(sp, sq, float_product):
    if float_product:
        g = (sp.astype(np.float32) * sq.astype(np.float32)).astype(np.float64)
    else:
        g = sp * sq
    return np.minimum(g, 90000.0)
