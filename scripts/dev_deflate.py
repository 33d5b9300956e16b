"""Development experiment (CPU, scipy): a global coarse space (exact lowest eigenvectors, or low-order polynomials of
the station position per degree of freedom) added to the additive hierarchy.  Not part of the product."""
import sys
src = open("scripts/dev_precond.py").read()
cut = src.index("# ---- bordered system with the 7 camera columns")
exec(src[:cut])
lv = build_hierarchy(Sc, 0.5, min_blocks=4, max_levels=3)
base = lambda x: apply_additive(lv, x)
run("additive 3 levels", base)

def with_space(Q):
    Q, _ = np.linalg.qr(Q)
    E = np.linalg.inv(Q.T @ (Sc @ Q))
    return lambda x: base(x) + Q @ (E @ (Q.T @ x))

# exact lowest eigenvectors of the preconditioned operator (ideal)
Mop = spla.LinearOperator((n, n), matvec=base)
for k in (8, 16, 32):
    X0 = rng.standard_normal((n, k))
    w, V = spla.lobpcg(Sc, X0, M=Mop, largest=False, tol=1e-5, maxiter=200)
    print("lowest eigenvalues of M S:", np.round(w[:8], 5))
    run("additive + %d exact eigenvectors" % k, with_space(V))

xy = net.eop[:, 3:5]
u = (xy - xy.mean(axis=0)) / xy.std(axis=0)
def poly_space(deg):
    terms = [(i, j) for i in range(deg + 1) for j in range(deg + 1 - i)]
    cols = []
    for (i, j) in terms:
        f = u[:, 0] ** i * u[:, 1] ** j
        for d in range(6):
            v = np.zeros((S, 6)); v[:, d] = f
            cols.append(v.ravel())
    return np.array(cols).T
for deg in (1, 2, 3):
    Q = poly_space(deg)
    run("additive + polynomials deg %d (%d vectors)" % (deg, Q.shape[1]), with_space(Q))
    # in physical variables (scaled variable = physical * (1 + column norm))
    Qp = Q * (1.0 + ns)[:, None]
    run("additive + physical polynomials deg %d" % deg, with_space(Qp))
