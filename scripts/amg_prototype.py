"""Study on the CPU: PCG iteration counts of the additive multilevel preconditioner against a multiplicative V-cycle on
the station part of a reduced camera system (scipy; the matrix is built from the oracle's Jacobian rows)."""
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

sys.path.insert(0, ".")
from bundleadjustment_amd import host  # noqa: E402
from oracle import py_oracle as po  # noqa: E402


def reduced_matrix(n_sta, pts_per_sta, rays=5, seed=3):
    hn = host.synth(n_cam=1, n_sta=n_sta, n_pts=n_sta * pts_per_sta, rays_per_point=rays, seed=seed, ap_free_mask=0, iop_free_mask=0)
    net = hn.net
    r, J = po.residual_jacobian_all(net)
    N, S, P = net.n_obs, net.n_sta, net.n_pts
    Js = J[:, :, 0:6]   # [N,2,6]
    Jp = J[:, :, 6:9]
    # Jacobi scaling 1 / (1 + column norm)
    cs = np.zeros((S, 6)); np.add.at(cs, net.obs_sta, np.sum(Js ** 2, axis=1))
    cp = np.zeros((P, 3)); np.add.at(cp, net.obs_pt, np.sum(Jp ** 2, axis=1))
    isig = 1.0 / net.pt_sigma
    cp += isig ** 2
    ss, sp_ = 1.0 / (1.0 + np.sqrt(cs)), 1.0 / (1.0 + np.sqrt(cp))
    Js = Js * ss[net.obs_sta][:, None, :]
    Jp = Jp * sp_[net.obs_pt][:, None, :]
    order = np.argsort(net.obs_pt, kind="stable")
    K = rays
    assert N == P * K
    Js = Js[order].reshape(P, K, 2, 6); Jp = Jp[order].reshape(P, K, 2, 3)
    sta = net.obs_sta[order].reshape(P, K)
    V = np.einsum("pkra,pkrb->pab", Jp, Jp) + np.einsum("pa,ab->pab", (isig * sp_) ** 2, np.eye(3))
    Vi = np.linalg.inv(V)
    W = np.einsum("pkra,pkrb->pkab", Js, Jp)          # [P,K,6,3]
    WV = np.einsum("pkab,pbc->pkac", W, Vi)
    blocks = -np.einsum("pkac,plbc->pklab", WV, W)     # [P,K,K,6,6]
    U = np.einsum("pkra,pkrb->pkab", Js, Js)
    for k in range(K):
        blocks[:, k, k] += U[:, k]
    rows = np.repeat(sta[:, :, None], K, axis=2).reshape(-1)
    cols = np.repeat(sta[:, None, :], K, axis=1).reshape(-1)
    # assemble BSR through a COO of block ids
    key = rows.astype(np.int64) * S + cols
    uniq, inv = np.unique(key, return_inverse=True)
    data = np.zeros((uniq.size, 6, 6)); np.add.at(data, inv, blocks.reshape(-1, 6, 6))
    br, bc = (uniq // S).astype(np.int32), (uniq % S).astype(np.int32)
    indptr = np.zeros(S + 1, dtype=np.int32); np.add.at(indptr, br + 1, 1); indptr = np.cumsum(indptr).astype(np.int32)
    A = sp.bsr_matrix((data, bc, indptr), shape=(6 * S, 6 * S))
    return A


def block_diag_inv(A):
    n = A.shape[0] // 6
    A = sp.bsr_matrix(A, blocksize=(6, 6))
    A.sort_indices()
    D = np.zeros((n, 6, 6))
    for i in range(n):
        for k in range(A.indptr[i], A.indptr[i + 1]):
            if A.indices[k] == i:
                D[i] = A.data[k]
    Di = np.linalg.inv(D)
    return sp.bsr_matrix((Di, np.arange(n, dtype=np.int32), np.arange(n + 1, dtype=np.int32)), shape=A.shape).tocsr()


def aggregate(A, theta):
    A = sp.bsr_matrix(A, blocksize=(6, 6)); A.sort_indices()
    n = A.shape[0] // 6
    fro = np.sqrt(np.sum(A.data ** 2, axis=(1, 2)))
    diag = np.zeros(n)
    rowof = np.repeat(np.arange(n), np.diff(A.indptr))
    diag[rowof[A.indices == rowof]] = fro[A.indices == rowof]
    sij = fro / np.sqrt(diag[rowof] * diag[A.indices])
    sij[A.indices == rowof] = 0
    rmax = np.zeros(n); np.maximum.at(rmax, rowof, sij)
    strong = (sij > 0) & (sij >= theta * rmax[rowof]) & (A.indices != rowof)
    nbrs = [A.indices[A.indptr[i]:A.indptr[i + 1]][strong[A.indptr[i]:A.indptr[i + 1]]] for i in range(n)]
    agg = -np.ones(n, dtype=np.int64); nc = 0
    for i in range(n):   # pass 1: roots whose strong neighbourhood is free
        if agg[i] < 0 and all(agg[j] < 0 for j in nbrs[i]):
            agg[i] = nc
            for j in nbrs[i]: agg[j] = nc
            nc += 1
    for i in range(n):   # pass 2: attach to a neighbouring aggregate
        if agg[i] < 0:
            for j in nbrs[i]:
                if agg[j] >= 0:
                    agg[i] = agg[j]; break
    for i in range(n):
        if agg[i] < 0:
            agg[i] = nc; nc += 1
    return agg, nc


def hierarchy(A, levels=3, theta=0.5):
    H = []
    for l in range(levels - 1):
        n = A.shape[0] // 6
        if n < 40: break
        Dinv = block_diag_inv(A)
        agg, nc = aggregate(A, theta)
        T = sp.kron(sp.csr_matrix((np.ones(n), (np.arange(n), agg)), shape=(n, nc)), sp.identity(6)).tocsr()
        DA = (Dinv @ A).tocsr()
        rho = abs(spla.eigs(DA, k=1, which="LM", return_eigenvectors=False, tol=1e-3)[0])
        omega = 4.0 / (3.0 * rho)
        Pm = (T - omega * (DA @ T)).tocsr()
        H.append(dict(A=A.tocsr(), Dinv=Dinv, P=Pm, rho=rho))
        A = sp.bsr_matrix((Pm.T @ A @ Pm), blocksize=(6, 6))
    Ac = A.tocsr()
    H.append(dict(A=Ac, Dinv=block_diag_inv(A), rho=abs(spla.eigs((block_diag_inv(A) @ Ac).tocsr(), k=1, which="LM", return_eigenvectors=False, tol=1e-3)[0])))
    return H


def additive(H, w=(0.5, 1.5)):
    def apply(r):
        z = H[0]["Dinv"] @ r
        if len(H) > 1:
            r1 = H[0]["P"].T @ r
            y1 = H[1]["Dinv"] @ r1
            if len(H) > 2:
                r2 = H[1]["P"].T @ r1
                y1 = y1 + w[1] * (H[1]["P"] @ (H[2]["Dinv"] @ r2))
            z = z + w[0] * (H[0]["P"] @ y1)
        return z
    return apply


def vcycle(H, nu=1, coarse="jacobi", omega_s=None):
    lu = spla.splu(H[-1]["A"].tocsc()) if coarse == "exact" else None

    def cyc(l, r):
        L = H[l]
        om = (omega_s if omega_s else 4.0 / 3.0) / L["rho"]
        if l == len(H) - 1:
            if lu is not None:
                return lu.solve(r)
            x = om * (L["Dinv"] @ r)
            for _ in range(2 * nu - 1):
                x = x + om * (L["Dinv"] @ (r - L["A"] @ x))
            return x
        x = om * (L["Dinv"] @ r)
        for _ in range(nu - 1):
            x = x + om * (L["Dinv"] @ (r - L["A"] @ x))
        rc = L["P"].T @ (r - L["A"] @ x)
        x = x + L["P"] @ cyc(l + 1, rc)
        for _ in range(nu):
            x = x + om * (L["Dinv"] @ (r - L["A"] @ x))
        return x
    return lambda r: cyc(0, r)


def pcg(A, b, M, rtol=1e-12, maxit=3000):
    x = np.zeros_like(b); r = b.copy(); z = M(r); p = z.copy(); rz = r @ z
    b0 = np.sqrt(rz)
    for it in range(1, maxit + 1):
        q = A @ p
        a = rz / (p @ q)
        x += a * p; r -= a * q
        z = M(r); rz2 = r @ z
        if np.sqrt(rz2) <= rtol * b0:
            return it
        p = z + (rz2 / rz) * p; rz = rz2
    return maxit


if __name__ == "__main__":
    n_sta = int(sys.argv[1]) if len(sys.argv) > 1 else 900
    pps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    t = time.time()
    A = reduced_matrix(n_sta, pps).tocsr()
    print("matrix", A.shape, "blocks/row", A.nnz / 36 / n_sta, "%.1fs" % (time.time() - t))
    rng = np.random.default_rng(1)
    b = rng.standard_normal(A.shape[0])
    H = hierarchy(sp.bsr_matrix(A, blocksize=(6, 6)))
    print("levels", [h["A"].shape[0] // 6 for h in H], "rho", [round(h["rho"], 2) for h in H])
    Dinv = H[0]["Dinv"]
    print("block-Jacobi      ", pcg(A, b, lambda r: Dinv @ r))
    print("additive (1,1)    ", pcg(A, b, additive(H, (1, 1))))
    print("additive (.5,1.5) ", pcg(A, b, additive(H)))
    for nu in (1, 2):
        for coarse in ("jacobi", "exact"):
            print("V(%d,%d) coarse=%s" % (nu, nu, coarse), pcg(A, b, vcycle(H, nu, coarse)))
    print("V(1,1) omega 1.0  ", pcg(A, b, vcycle(H, 1, "jacobi", 1.0)))


def candidates(A, Dinv, m=6, sweeps=40, kind="relax", seed=0):
    n = A.shape[0]
    if kind == "eig":
        Dh = sp.bsr_matrix(Dinv, blocksize=(6, 6))
        # generalized problem A v = lambda D v  via  shift-invert free: use lobpcg with the preconditioner
        X = np.random.default_rng(seed).standard_normal((n, m))
        D = block_diag_inv(sp.bsr_matrix(Dinv, blocksize=(6, 6)))
        w, V = spla.lobpcg(A, X, B=D, M=Dinv, largest=False, tol=1e-6, maxiter=400)
        print("   lowest eigenvalues of D^-1 A:", np.round(w, 6))
        return V
    X = np.random.default_rng(seed).standard_normal((n, m))
    DA = (Dinv @ A).tocsr()
    for _ in range(sweeps):
        X = X - 0.6 * (DA @ X)
        X /= np.linalg.norm(X, axis=0)
    return X


def hierarchy_B(A, B, levels=3, theta=0.5):
    H = []
    for l in range(levels - 1):
        n = A.shape[0] // 6
        if n < 40: break
        Dinv = block_diag_inv(A)
        agg, nc = aggregate(A, theta)
        m = B.shape[1]
        Tb = np.zeros((n, 6, m)); Bc = np.zeros((nc * m, m))
        for c in range(nc):
            idx = np.nonzero(agg == c)[0]
            Bl = B.reshape(n, 6, m)[idx].reshape(-1, m)
            if Bl.shape[0] >= m:
                Q, R = np.linalg.qr(Bl)
            else:  # aggregate too small for m candidates: pad
                Q = np.zeros((Bl.shape[0], m)); Q[:, :Bl.shape[0]] = np.eye(Bl.shape[0]); R = np.zeros((m, m)); R[:Bl.shape[0]] = Bl
            Tb[idx] = Q.reshape(len(idx), 6, m)
            Bc[c * m:(c + 1) * m] = R
        T = sp.bsr_matrix((Tb, agg.astype(np.int32), np.arange(n + 1, dtype=np.int32)), shape=(6 * n, m * nc)).tocsr()
        DA = (Dinv @ A).tocsr()
        rho = abs(spla.eigs(DA, k=1, which="LM", return_eigenvectors=False, tol=1e-3)[0])
        Pm = (T - (4.0 / (3.0 * rho)) * (DA @ T)).tocsr()
        H.append(dict(A=A.tocsr(), Dinv=Dinv, P=Pm, rho=rho))
        A = sp.bsr_matrix((Pm.T @ A @ Pm), blocksize=(6, 6))
        B = Bc
    Ac = A.tocsr()
    Di = block_diag_inv(A)
    H.append(dict(A=Ac, Dinv=Di, rho=abs(spla.eigs((Di @ Ac).tocsr(), k=1, which="LM", return_eigenvectors=False, tol=1e-3)[0])))
    return H


if __name__ == "__main__":
    for kind in ("relax", "eig"):
        B = candidates(A, Dinv, kind=kind)
        HB = hierarchy_B(sp.bsr_matrix(A, blocksize=(6, 6)), B)
        print(kind, "levels", [h["A"].shape[0] // 6 for h in HB])
        print("  additive (1,1)   ", pcg(A, b, additive(HB, (1, 1))))
        print("  additive (.5,1.5)", pcg(A, b, additive(HB)))
        print("  V(1,1) jacobi    ", pcg(A, b, vcycle(HB, 1, "jacobi")))
        print("  V(1,1) exact     ", pcg(A, b, vcycle(HB, 1, "exact")))


def agg_block_inverse(A, agg, nc):
    """Non-overlapping additive Schwarz: inverse of the diagonal blocks of A over the aggregates."""
    A = A.tocsr()
    n = A.shape[0] // 6
    blocks = []
    perm = []
    for c in range(nc):
        idx = np.nonzero(agg == c)[0]
        dof = (6 * idx[:, None] + np.arange(6)[None, :]).reshape(-1)
        blocks.append(np.linalg.inv(A[dof][:, dof].toarray()))
        perm.append(dof)
    perm = np.concatenate(perm)
    Bd = sp.block_diag(blocks, format="csr")
    Pm = sp.csr_matrix((np.ones(perm.size), (perm, np.arange(perm.size))), shape=(6 * n, 6 * n))
    return (Pm @ Bd @ Pm.T).tocsr()


if __name__ == "__main__":
    agg0, nc0 = aggregate(sp.bsr_matrix(A, blocksize=(6, 6)), 0.5)
    # a second, shifted partition for the smoother so that its block boundaries differ from the aggregates of P0
    Bi = agg_block_inverse(A, agg0, nc0)
    H2 = [dict(h) for h in H]
    H2[0]["Dinv"] = Bi
    H2[0]["rho"] = abs(spla.eigs((Bi @ A).tocsr(), k=1, which="LM", return_eigenvectors=False, tol=1e-3)[0])
    print("aggregate-block smoother at level 0, rho", round(H2[0]["rho"], 2))
    print("  block only       ", pcg(A, b, lambda r: Bi @ r))
    print("  additive (1,1)   ", pcg(A, b, additive(H2, (1, 1))))
    print("  additive (.5,1.5)", pcg(A, b, additive(H2)))
    print("  V(1,1) jacobi    ", pcg(A, b, vcycle(H2, 1, "jacobi")))


def deflated_pcg(A, b, M, W, rtol=1e-12, maxit=3000):
    """PCG on the complement of span(W) (A-orthogonal projection), exact coarse solve on W."""
    AW = A @ W
    E = np.linalg.inv(W.T @ AW)
    def proj(v):   # P v = v - A W E W^T v
        return v - AW @ (E @ (W.T @ v))
    x = W @ (E @ (W.T @ b))
    r = proj(b - A @ x)
    z = M(r); p = z.copy(); rz = r @ z
    b0 = np.sqrt(b @ M(b))
    for it in range(1, maxit + 1):
        q = proj(A @ p)
        a = rz / (p @ q)
        x += a * p; r -= a * q
        z = M(r); rz2 = r @ z
        if np.sqrt(rz2) <= rtol * b0:
            return it
        p = z + (rz2 / rz) * p; rz = rz2
    return maxit


if __name__ == "__main__":
    Madd = additive(H)
    n = A.shape[0]
    Mop = spla.LinearOperator((n, n), matvec=Madd, matmat=lambda X: np.column_stack([Madd(X[:, k]) for k in range(X.shape[1])]))
    X = np.random.default_rng(5).standard_normal((n, 48))
    w, V = spla.lobpcg(A, X, M=Mop, largest=False, tol=1e-5, maxiter=300)   # Rayleigh quotients of A with M as preconditioner: not M^-1 A's spectrum, but its low modes
    print("lowest Ritz values of A:", np.round(w[:8], 6))
    for k in (0, 8, 16, 32, 48):
        print("  deflating", k, "->", deflated_pcg(A, b, Madd, V[:, :k]) if k else pcg(A, b, Madd))
