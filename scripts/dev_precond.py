"""Development experiment (CPU, scipy): PCG iteration counts on the reduced camera system of the
synthetic cfg4 network for candidate preconditioners.  Not part of the product or of the tests."""
import sys, time
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
import scipy.linalg as sla
sys.path.insert(0, ".")
from bundleadjustment_amd import host
from oracle import py_oracle as orc

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
hn = host.synth_config("cfg4", scale)
net = hn.net
N, S, P = net.n_obs, net.n_sta, net.n_pts
t0 = time.time()
r, J = orc.residual_jacobian_all(net)
print("N S P", N, S, P, "jac %.1fs" % (time.time() - t0), flush=True)
Js = J[:, :, 0:6]; Jp = J[:, :, 6:9]; Jcam = J[:, :, [9, 10, 11, 14, 15, 17, 18]]
rows = (2 * np.arange(N)[:, None, None] + np.arange(2)[None, :, None])
cs = (6 * net.obs_sta.astype(np.int64)[:, None, None] + np.arange(6)[None, None, :])
cp = (3 * net.obs_pt.astype(np.int64)[:, None, None] + np.arange(3)[None, None, :])
A_s = sp.csr_matrix((Js.ravel(), (np.broadcast_to(rows, Js.shape).ravel(), np.broadcast_to(cs, Js.shape).ravel())), shape=(2 * N, 6 * S))
A_p = sp.csr_matrix((Jp.ravel(), (np.broadcast_to(rows, Jp.shape).ravel(), np.broadcast_to(cp, Jp.shape).ravel())), shape=(2 * N, 3 * P))
A_c = Jcam.reshape(2 * N, 7).copy()
del J, Js, Jp
isig = (1.0 / net.pt_sigma).ravel()
# Jacobi scaling as the solver does it: 1 / (1 + column norm)
ns = np.sqrt(np.asarray(A_s.multiply(A_s).sum(axis=0)).ravel())
npn = np.sqrt(np.asarray(A_p.multiply(A_p).sum(axis=0)).ravel() + isig ** 2)
A_s = A_s @ sp.diags(1.0 / (1.0 + ns)); A_p = A_p @ sp.diags(1.0 / (1.0 + npn))
A_c = A_c / (1.0 + np.sqrt((A_c ** 2).sum(axis=0)))
pri = isig / (1.0 + npn)
lm = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-4
U = (A_s.T @ A_s).tocsr()
U = U + sp.diags(lm * U.diagonal())
V = (A_p.T @ A_p).tocsr() + sp.diags(pri ** 2)
V = V + sp.diags(lm * V.diagonal())
W = (A_s.T @ A_p).tocsr()
Vb = V.tobsr(blocksize=(3, 3))
Vinv = sp.bsr_matrix((np.linalg.inv(Vb.data), Vb.indices, Vb.indptr), shape=V.shape)
t0 = time.time()
Sm = (U - W @ (Vinv @ W.T)).tobsr(blocksize=(6, 6))
print("reduced system built %.1fs, blocks %d (%.1f per row)" % (time.time() - t0, Sm.data.shape[0], Sm.data.shape[0] / S), flush=True)
Sm.sort_indices()
Sc = Sm.tocsr()
n = 6 * S
rng = np.random.default_rng(0)
b = Sc @ rng.standard_normal(n)

D = Sm.diagonal()
# block-Jacobi
dblk = np.zeros((S, 6, 6))
for i in range(S):
    lo, hi = Sm.indptr[i], Sm.indptr[i + 1]
    k = lo + np.searchsorted(Sm.indices[lo:hi], i)
    dblk[i] = Sm.data[k]
dinv = np.linalg.inv(dblk)
def bj(x):
    return np.einsum("sij,sj->si", dinv, x.reshape(S, 6)).ravel()

def run(name, M, tol=1e-10, maxit=5000):
    it = [0]
    def cb(xk): it[0] += 1
    t0 = time.time()
    x, info = spla.cg(Sc, b, rtol=tol, maxiter=maxit, M=spla.LinearOperator((n, n), matvec=M), callback=cb)
    print("%-46s iters %5d  info %d  true relres %.2e  (%.1fs)" % (name, it[0], info, np.linalg.norm(b - Sc @ x) / np.linalg.norm(b), time.time() - t0), flush=True)
    return it[0]

run("block-Jacobi", bj)

# station positions -> geometric aggregates of about a x a stations
XY = net.eop[:, 3:5]
area = (XY[:, 0].max() - XY[:, 0].min()) * (XY[:, 1].max() - XY[:, 1].min())
h = np.sqrt(area / S)
def aggregates_geo(a):
    gx = np.floor((XY[:, 0] - XY[:, 0].min()) / (a * h) + 1e-9).astype(int)
    gy = np.floor((XY[:, 1] - XY[:, 1].min()) / (a * h) + 1e-9).astype(int)
    key = gx * (gy.max() + 1) + gy
    _, agg = np.unique(key, return_inverse=True)
    return agg

def tentative(agg, nfine):
    rowsT = np.arange(6 * nfine)
    colsT = 6 * agg[rowsT // 6] + rowsT % 6
    return sp.csr_matrix((np.ones(6 * nfine), (rowsT, colsT)), shape=(6 * nfine, 6 * (agg.max() + 1)))

def blockdiag_inv(A, nb):
    B = A.tobsr(blocksize=(6, 6)); B.sort_indices()
    out = np.zeros((nb, 6, 6))
    for i in range(nb):
        lo, hi = B.indptr[i], B.indptr[i + 1]
        out[i] = B.data[lo + np.searchsorted(B.indices[lo:hi], i)]
    inv = np.linalg.inv(out)
    return lambda x: np.einsum("sij,sj->si", inv, x.reshape(nb, 6)).ravel()

def coarse_solver(Ac):
    c = sla.cho_factor(Ac.toarray())
    return lambda y: sla.cho_solve(c, y)

Dinv_bsr = sp.bsr_matrix((dinv, np.arange(S), np.arange(S + 1)), shape=(n, n))
def smooth_prolongator(T, omega):
    return (T - omega * (Dinv_bsr @ (Sc @ T))).tocsr()
# spectral radius estimate of Dinv S for omega
v = rng.standard_normal(n)
for _ in range(30):
    v = bj(Sc @ v); lam = np.linalg.norm(v); v /= lam
print("rho(Dinv S) ~ %.3f" % lam)
def bsr_sorted(A):
    B = A.tobsr(blocksize=(6, 6)); B.sort_indices(); return B

def strength_graph(B):
    nb = B.shape[0] // 6
    fro = np.sqrt((B.data ** 2).sum(axis=(1, 2)))
    rowi = np.repeat(np.arange(nb), np.diff(B.indptr))
    dg = np.zeros(nb); m = rowi == B.indices; dg[rowi[m]] = fro[m]
    sij = fro / np.sqrt(dg[rowi] * dg[B.indices])
    sij[m] = 0.0
    return rowi, B.indices, sij, B.indptr

def aggregate_graph(B, theta):
    nb = B.shape[0] // 6
    rowi, col, sij, ptr = strength_graph(B)
    rmax = np.maximum.reduceat(sij, ptr[:-1])
    strong = sij >= theta * rmax[rowi]
    strong &= (col != rowi)
    agg = -np.ones(nb, dtype=np.int64)
    na = 0
    for i in range(nb):
        if agg[i] >= 0: continue
        nb_ = col[ptr[i]:ptr[i + 1]][strong[ptr[i]:ptr[i + 1]]]
        if np.all(agg[nb_] < 0):
            agg[i] = na; agg[nb_] = na; na += 1
    agg1 = agg.copy()
    for i in range(nb):
        if agg[i] >= 0: continue
        sl = slice(ptr[i], ptr[i + 1])
        c, sv, st = col[sl], sij[sl], strong[sl]
        ok = st & (agg1[c] >= 0)
        if ok.any():
            agg[i] = agg1[c[ok][np.argmax(sv[ok])]]
    for i in range(nb):
        if agg[i] >= 0: continue
        sl = slice(ptr[i], ptr[i + 1])
        c, st = col[sl], strong[sl]
        agg[i] = na
        for j in c[st]:
            if agg[j] < 0: agg[j] = na
        na += 1
    return agg, na

def build_hierarchy(A0, theta, min_blocks=1, smooth=True, max_levels=8):
    """levels: list of dict(P, dinv) ; additive application."""
    levels = []
    A = A0
    while True:
        B = bsr_sorted(A)
        nb = B.shape[0] // 6
        d = np.zeros((nb, 6, 6))
        rowi = np.repeat(np.arange(nb), np.diff(B.indptr))
        m = rowi == B.indices
        d[rowi[m]] = B.data[m]
        di = np.linalg.inv(d)
        lev = dict(nb=nb, dinv=di, A=A, blocks=B.data.shape[0])
        levels.append(lev)
        if nb <= min_blocks or len(levels) >= max_levels: break
        agg, na = aggregate_graph(B, theta)
        if na >= nb: break
        T = tentative(agg, nb)
        if smooth:
            Dinv = sp.bsr_matrix((di, np.arange(nb), np.arange(nb + 1)), shape=A.shape)
            jac = lambda x, di=di, nb=nb: np.einsum("sij,sj->si", di, x.reshape(nb, 6)).ravel()
            l = power_rho(A, jac, 6 * nb)
            P = (T - (4.0 / (3.0 * l)) * (Dinv @ (A @ T))).tocsr()
        else:
            P = T
        lev["P"] = P
        lev["na"] = na
        A = (P.T @ A @ P).tocsr()
    return levels

def power_rho(A, inv, m):
    v = rng.standard_normal(m)
    for _ in range(30):
        v = inv(A @ v); l = np.linalg.norm(v); v /= l
    return l

def apply_additive(levels, x, exact_last=None):
    rs = [x]
    for lev in levels[:-1]:
        rs.append(lev["P"].T @ rs[-1])
    y = None
    for k in range(len(levels) - 1, -1, -1):
        lev = levels[k]
        if k == len(levels) - 1 and exact_last is not None:
            yk = exact_last(rs[k])
        else:
            yk = np.einsum("sij,sj->si", lev["dinv"], rs[k].reshape(lev["nb"], 6)).ravel()
        if y is not None: yk = yk + lev["P"] @ y
        y = yk
    return y

def apply_weighted(levels, x, wts):
    rs = [x]
    for lev in levels[:-1]:
        rs.append(lev["P"].T @ rs[-1])
    y = None
    for k in range(len(levels) - 1, -1, -1):
        lev = levels[k]
        yk = wts[min(k, len(wts) - 1)] * np.einsum("sij,sj->si", lev["dinv"], rs[k].reshape(lev["nb"], 6)).ravel()
        if y is not None: yk = yk + lev["P"] @ y
        y = yk
    return y
# ---- bordered system with the 7 camera columns
ApTAc = A_p.T @ A_c                      # 3P x 7
T_ = Vinv @ ApTAc
Ssc = A_s.T @ A_c - W @ T_               # 6S x 7
Scc = A_c.T @ A_c - ApTAc.T @ T_
Scc = Scc + np.diag(lm * np.diag(Scc))
nfull = n + 7
def Afull(x):
    xs, xc = x[:n], x[n:]
    return np.concatenate([Sc @ xs + Ssc @ xc, Ssc.T @ xs + Scc @ xc])
Aop = spla.LinearOperator((nfull, nfull), matvec=Afull)
bfull = Afull(rng.standard_normal(nfull))
def runfull(name, M, tol=1e-10, maxit=5000):
    it = [0]
    def cb(xk): it[0] += 1
    x, info = spla.cg(Aop, bfull, rtol=tol, maxiter=maxit, M=spla.LinearOperator((nfull, nfull), matvec=M), callback=cb)
    print("%-60s iters %5d  relres %.2e" % (name, it[0], np.linalg.norm(bfull - Afull(x)) / np.linalg.norm(bfull)), flush=True)
levels = build_hierarchy(Sc, 0.5)
lv = levels[:3]
Mss = lambda x: apply_additive(lv, x)
Scc_inv = np.linalg.inv(Scc)
run("stations only, 3-level additive", Mss)
runfull("bordered, block-diagonal [AMG, Scc]", lambda x: np.concatenate([Mss(x[:n]), Scc_inv @ x[n:]]))
# arrow preconditioner with G ~ Sss^-1 Ssc
def arrow(G, scale=1.0):
    Shat = Scc - Ssc.T @ G
    w = np.linalg.eigvalsh(Shat)
    print("   eig(Shat) min %.3e max %.3e" % (w.min(), w.max()))
    Si = np.linalg.inv(Shat)
    def M(x):
        ts = Mss(x[:n]) * scale
        tc = x[n:] - G.T @ x[:n]
        zc = Si @ tc
        return np.concatenate([ts - G @ zc, zc])
    return M
def pcg_cols(A, M, k_it, x0=None):
    cols = []
    for k in range(7):
        x, info = spla.cg(A, Ssc[:, k], x0=None if x0 is None else x0[:, k], rtol=1e-30, maxiter=k_it, M=spla.LinearOperator((n, n), matvec=M))
        cols.append(x)
    return np.column_stack(cols)
def coarse(G):
    # z = [M^-1 r_s; 0] + Q (Q^T S Q)^-1 Q^T r,  Q = [-G; I]
    SG = Sc @ G
    E = Scc - Ssc.T @ G - G.T @ Ssc + G.T @ SG
    Ei = np.linalg.inv(E)
    def M(x):
        zc = Ei @ (x[n:] - G.T @ x[:n])
        return np.concatenate([Mss(x[:n]) - G @ zc, zc])
    return M
for k_it in (3, 10):
    runfull("coarse space Q=[-G;I], G from %d PCG iterations" % k_it, coarse(pcg_cols(Sc, Mss, k_it)))
# stale G: computed for a more strongly damped matrix (earlier LM iteration)
U0 = (A_s.T @ A_s).tocsr()
for lm_old in (1e-4, 1e-5):
    S_old = (Sc + sp.diags((lm_old - lm) * U0.diagonal())).tocsr()
    lev_old = build_hierarchy(S_old, 0.5)[:3]
    G_old = pcg_cols(S_old, lambda x: apply_additive(lev_old, x), 10)
    runfull("coarse space with G from the matrix damped %.0e (10 its)" % lm_old, coarse(G_old))
    runfull("   same, refreshed by 2 warm-started PCG iterations", coarse(pcg_cols(Sc, Mss, 2, G_old)))
sys.exit(0)
def nested(aggf, a2):
    cx = np.bincount(aggf, XY[:, 0]) / np.bincount(aggf); cy = np.bincount(aggf, XY[:, 1]) / np.bincount(aggf)
    gx = np.floor((cx - XY[:, 0].min()) / (a2 * h) + 1e-9).astype(int); gy = np.floor((cy - XY[:, 1].min()) / (a2 * h) + 1e-9).astype(int)
    _, agg2 = np.unique(gx * (gy.max() + 1) + gy, return_inverse=True)
    return agg2

def power_rho(A, inv, m):
    v = rng.standard_normal(m)
    for _ in range(30):
        v = inv(A @ v); l = np.linalg.norm(v); v /= l
    return l

for a1, a2 in ((3, 9), (3, 12), (2, 6)):
    agg1 = aggregates_geo(a1); na1 = agg1.max() + 1
    T1 = tentative(agg1, S)
    P1 = smooth_prolongator(T1, 4.0 / (3.0 * lam))
    A1 = (P1.T @ Sc @ P1).tocsr()
    j1 = blockdiag_inv(A1, na1)
    c1 = coarse_solver(A1)
    run("additive smoothed 2-level a=%d exact coarse (%d dof)" % (a1, 6 * na1), lambda x: bj(x) + P1 @ c1(P1.T @ x))
    agg2 = nested(agg1, a2); na2 = agg2.max() + 1
    T2 = tentative(agg2, na1)
    l1 = power_rho(A1, j1, 6 * na1)
    B1 = A1.tobsr(blocksize=(6, 6)); B1.sort_indices()
    d1 = np.zeros((na1, 6, 6))
    for i in range(na1):
        lo, hi = B1.indptr[i], B1.indptr[i + 1]
        d1[i] = B1.data[lo + np.searchsorted(B1.indices[lo:hi], i)]
    D1inv = sp.bsr_matrix((np.linalg.inv(d1), np.arange(na1), np.arange(na1 + 1)), shape=A1.shape)
    P2 = (T2 - (4.0 / (3.0 * l1)) * (D1inv @ (A1 @ T2))).tocsr()
    A2 = (P2.T @ A1 @ P2).tocsr()
    c2 = coarse_solver(A2)
    P12 = (P1 @ P2).tocsr()
    print("   levels: %d -> %d -> %d dof; P1 blocks/row %.1f, A1 blocks/row %.1f, P2 blocks/row %.1f" % (n, 6 * na1, 6 * na2, P1.nnz / 36 / S, A1.nnz / 36 / na1, P2.nnz / 36 / na1))
    run("additive smoothed 3-level a=%d/%d" % (a1, a2), lambda x: bj(x) + P1 @ (j1(P1.T @ x) + P2 @ c2(P2.T @ (P1.T @ x))))
    def v3(x, w=0.7):
        # V(1,1) on the fine level; level 1 handled by additive (block-Jacobi + exact level 2)
        z = w * bj(x)
        rr = P1.T @ (x - Sc @ z)
        z = z + P1 @ (j1(rr) + P2 @ c2(P2.T @ rr))
        return z + w * bj(x - Sc @ z)
    run("fine V(1,1) + additive levels 1,2 a=%d/%d" % (a1, a2), v3)
    def v3full(x, w=0.7):
        z = w * bj(x)
        r1 = P1.T @ (x - Sc @ z)
        y1 = w * j1(r1)
        y1 = y1 + P2 @ c2(P2.T @ (r1 - A1 @ y1))
        y1 = y1 + w * j1(r1 - A1 @ y1)
        z = z + P1 @ y1
        return z + w * bj(x - Sc @ z)
    run("full V(1,1) 3-level a=%d/%d" % (a1, a2), v3full)
