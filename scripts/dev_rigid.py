"""Development experiment (CPU, scipy): smoothed aggregation whose tentative prolongator carries the six rigid-body
modes of an aggregate (instead of six constants), on the reduced camera system of cfg4.  Not part of the product."""
import sys
src = open("scripts/dev_precond.py").read()
cut = src.index("# ---- bordered system with the 7 camera columns")
exec(src[:cut])

def rot(a):
    so, co, sp, cp, sk, ck = np.sin(a[0]), np.cos(a[0]), np.sin(a[1]), np.cos(a[1]), np.sin(a[2]), np.cos(a[2])
    return np.array([[cp * ck, co * sk + so * sp * ck, so * sk - co * sp * ck],
                     [-cp * sk, co * ck - so * sp * sk, so * ck + co * sp * sk],
                     [sp, -so * cp, co * cp]])
def skew(d):
    return np.array([[0, -d[2], d[1]], [d[2], 0, -d[0]], [-d[1], d[0], 0]])
# angle rates for a world rotation delta: dR = -R [delta]x
M = np.zeros((S, 3, 3))
for i in range(S):
    a = net.eop[i, :3]
    R = rot(a)
    h = 1e-6
    dR = np.stack([(rot(a + h * np.eye(3)[k]) - rot(a - h * np.eye(3)[k])).ravel() / (2 * h) for k in range(3)], axis=1)
    rhs = np.stack([(-R @ skew(np.eye(3)[j])).ravel() for j in range(3)], axis=1)
    M[i] = np.linalg.lstsq(dR, rhs, rcond=None)[0]
pos = net.eop[:, 3:6]
colscale = (1.0 + ns).reshape(S, 6)   # scaled variable = physical * (1 + column norm)

def tentative_rigid(agg, nfine, pos_f, M_f, scale_f):
    na = agg.max() + 1
    cen = np.zeros((na, 3)); cnt = np.zeros(na)
    np.add.at(cen, agg, pos_f); np.add.at(cnt, agg, 1.0)
    cen /= cnt[:, None]
    blocks = np.zeros((nfine, 6, 6))
    for i in range(nfine):
        T = np.zeros((6, 6))
        T[:3, :3] = M_f[i] if M_f is not None else np.eye(3)
        T[3:, :3] = -skew(pos_f[i] - cen[agg[i]])
        T[3:, 3:] = np.eye(3)
        if scale_f is not None:
            T = scale_f[i][:, None] * T
        blocks[i] = T
    return sp.bsr_matrix((blocks, agg.astype(np.int32), np.arange(nfine + 1, dtype=np.int32)), shape=(6 * nfine, 6 * na)).tocsr(), cen

def build_rigid(A0, theta, max_levels, rigid=True, normalize=True):
    levels = []
    A = A0; pos_f = pos; M_f = M; scale_f = colscale
    while True:
        B = bsr_sorted(A)
        nb = B.shape[0] // 6
        d = np.zeros((nb, 6, 6))
        rowi = np.repeat(np.arange(nb), np.diff(B.indptr))
        m = rowi == B.indices
        d[rowi[m]] = B.data[m]
        di = np.linalg.inv(d)
        lev = dict(nb=nb, dinv=di, A=A)
        levels.append(lev)
        if nb <= 4 or len(levels) >= max_levels: break
        agg, na = aggregate_graph(B, theta)
        if na >= nb: break
        if rigid:
            T, cen = tentative_rigid(agg, nb, pos_f, M_f, scale_f)
            if normalize:
                # orthonormalise the columns of each aggregate's block (QR per aggregate), as SA does
                Tb = T.tobsr(blocksize=(6, 6))
                data = Tb.data.copy()
                for a_ in range(na):
                    rows_ = np.where(agg == a_)[0]
                    Q, Rr = np.linalg.qr(data[rows_].reshape(-1, 6))
                    data[rows_] = Q.reshape(-1, 6, 6)
                T = sp.bsr_matrix((data, Tb.indices, Tb.indptr), shape=T.shape).tocsr()
        else:
            T = tentative(agg, nb); cen = None
        Dinv = sp.bsr_matrix((di, np.arange(nb), np.arange(nb + 1)), shape=A.shape)
        jac = lambda x, di=di, nb=nb: np.einsum("sij,sj->si", di, x.reshape(nb, 6)).ravel()
        l = power_rho(A, jac, 6 * nb)
        P = (T - (4.0 / (3.0 * l)) * (Dinv @ (A @ T))).tocsr()
        lev["P"] = P
        A = (P.T @ A @ P).tocsr()
        if rigid:
            pos_f = cen; M_f = None; scale_f = None
    return levels

for ml in (3, 4):
    lv = build_hierarchy(Sc, 0.5, min_blocks=4, max_levels=ml)
    run("constants, additive, %d levels %s" % (len(lv), [l["nb"] for l in lv]), lambda x: apply_additive(lv, x))
    for norm in (False, True):
        lr = build_rigid(Sc, 0.5, ml, True, norm)
        run("rigid modes (qr=%d), additive, %d levels %s" % (norm, len(lr), [l["nb"] for l in lr]), lambda x: apply_additive(lr, x))
