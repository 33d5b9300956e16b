"""Development experiment (CPU, scipy): multiplicative V-cycle vs the additive hierarchy as PCG preconditioner
on the reduced camera system of cfg4.  Not part of the product or of the tests."""
import sys
src = open("scripts/dev_precond.py").read()
cut = src.index("# ---- bordered system with the 7 camera columns")
exec(src[:cut])

def jac_of(lev):
    return lambda x: np.einsum("sij,sj->si", lev["dinv"], x.reshape(lev["nb"], 6)).ravel()

def vcycle(levels, k, r, w, nu=1, exact_last=None):
    lev = levels[k]
    J = jac_of(lev)
    if k == len(levels) - 1:
        if exact_last is not None:
            return exact_last(r)
        return J(r) * w[k]
    A = lev["A"]
    z = w[k] * J(r)
    for _ in range(nu - 1):
        z = z + w[k] * J(r - A @ z)
    rc = lev["P"].T @ (r - A @ z)
    z = z + lev["P"] @ vcycle(levels, k + 1, rc, w, nu, exact_last)
    for _ in range(nu):
        z = z + w[k] * J(r - A @ z)
    return z

for theta in (0.5,):
    for ml in (3, 4, 6):
        lv = build_hierarchy(Sc, theta, min_blocks=4, max_levels=ml)
        print("theta %.2f levels %s" % (theta, [l["nb"] for l in lv]), flush=True)
        run("additive %d levels" % len(lv), lambda x: apply_additive(lv, x))
        # smoother weights: 1/rho(Dinv A) per level
        w = []
        for l in lv:
            w.append(1.0 / power_rho(l["A"], jac_of(l), 6 * l["nb"]))
        print("   1/rho per level", np.round(w, 3))
        for f in (1.0, 4.0 / 3.0):
            ww = [f * x for x in w]
            run("V(1,1) omega=%.2f/rho, %d levels, Jacobi last" % (f, len(lv)), lambda x: vcycle(lv, 0, x, ww))
        ex = coarse_solver(lv[-1]["A"]) if lv[-1]["nb"] <= 400 else None
        if ex is not None:
            run("V(1,1) omega=1/rho, exact last", lambda x: vcycle(lv, 0, x, w, 1, ex))
            run("additive, exact last", lambda x: apply_additive(lv, x, ex))
