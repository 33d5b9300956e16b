"""Development experiment (CPU, scipy): two damped-Jacobi smoothing steps in the prolongator.  Not part of the product."""
import sys
src = open("scripts/dev_precond.py").read()
cut = src.index("# ---- bordered system with the 7 camera columns")
exec(src[:cut])
def build(A0, theta, max_levels, nsmooth, omega_f=4.0 / 3.0):
    levels = []; A = A0
    while True:
        B = bsr_sorted(A); nb = B.shape[0] // 6
        d = np.zeros((nb, 6, 6)); rowi = np.repeat(np.arange(nb), np.diff(B.indptr)); m = rowi == B.indices
        d[rowi[m]] = B.data[m]; di = np.linalg.inv(d)
        lev = dict(nb=nb, dinv=di, A=A); levels.append(lev)
        if nb <= 4 or len(levels) >= max_levels: break
        agg, na = aggregate_graph(B, theta)
        if na >= nb: break
        T = tentative(agg, nb)
        Dinv = sp.bsr_matrix((di, np.arange(nb), np.arange(nb + 1)), shape=A.shape)
        jac = lambda x, di=di, nb=nb: np.einsum("sij,sj->si", di, x.reshape(nb, 6)).ravel()
        l = power_rho(A, jac, 6 * nb)
        P = T
        for _ in range(nsmooth):
            P = (P - (omega_f / l) * (Dinv @ (A @ P))).tocsr()
        lev["P"] = P
        A = (P.T @ A @ P).tocsr()
        lev["Pnnz"] = P.nnz / 36
    return levels
for ns in (1, 2, 3):
    for of in (4.0 / 3.0,):
        lv = build(Sc, 0.5, 3, ns, of)
        run("smoothing steps %d, omega %.2f/rho, levels %s, P0 blocks/row %.1f" % (ns, of, [l["nb"] for l in lv], lv[0]["Pnnz"] / S), lambda x: apply_additive(lv, x))
