"""Development experiment (CPU, scipy): how much a better solve on level 1 would buy the additive scheme
(exact A1^-1, or a multiplicative cycle over levels 1..L).  Not part of the product."""
import sys
src = open("scripts/dev_precond.py").read()
cut = src.index("# ---- bordered system with the 7 camera columns")
exec(src[:cut])
lv = build_hierarchy(Sc, 0.5, min_blocks=4, max_levels=3)
run("additive 3 levels", lambda x: apply_additive(lv, x))
A1 = lv[1]["A"].tocsc()
lu = spla.splu(A1)
P0 = lv[0]["P"]
d0 = lambda x: np.einsum("sij,sj->si", lv[0]["dinv"], x.reshape(lv[0]["nb"], 6)).ravel()
run("fine Jacobi + exact level-1 solve (additive)", lambda x: d0(x) + P0 @ lu.solve(P0.T @ x))
def jac_of(lev):
    return lambda x: np.einsum("sij,sj->si", lev["dinv"], x.reshape(lev["nb"], 6)).ravel()
def vcyc(k, r, w):
    lev = lv[k]; J = jac_of(lev)
    if k == len(lv) - 1: return w[k] * J(r)
    A = lev["A"]
    z = w[k] * J(r)
    z = z + lev["P"] @ vcyc(k + 1, lev["P"].T @ (r - A @ z), w)
    return z + w[k] * J(r - A @ z)
w = [1.0 / power_rho(l["A"], jac_of(l), 6 * l["nb"]) for l in lv]
run("fine Jacobi + V(1,1) over levels 1..2 (additive at the top)", lambda x: d0(x) + P0 @ vcyc(1, P0.T @ x, w))
lv4 = build_hierarchy(Sc, 0.5, min_blocks=4, max_levels=5)
run("additive %d levels" % len(lv4), lambda x: apply_additive(lv4, x))
