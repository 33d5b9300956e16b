"""Development experiment (CPU, scipy): the fine-level term D0^-1 of the additive hierarchy replaced by the exact
inverse of the diagonal block of each aggregate (non-overlapping additive Schwarz).  Not part of the product."""
import sys
src = open("scripts/dev_precond.py").read()
cut = src.index("# ---- bordered system with the 7 camera columns")
exec(src[:cut])
lv = build_hierarchy(Sc, 0.5, min_blocks=4, max_levels=3)
run("additive 3 levels (6x6 blocks on the fine level)", lambda x: apply_additive(lv, x))
B0 = bsr_sorted(Sc)
def agg_block_solver(agg, na):
    idx = [np.where(agg == a)[0] for a in range(na)]
    dof = [np.concatenate([6 * i + np.arange(6) for i in ii]) for ii in idx]
    inv = [np.linalg.inv(Sc[d][:, d].toarray()) for d in dof]
    def apply(x):
        y = np.zeros_like(x)
        for d, m in zip(dof, inv):
            y[d] = m @ x[d]
        return y
    return apply, np.mean([len(d) for d in dof])
for theta in (0.5, 0.25):
    agg, na = aggregate_graph(B0, theta)
    solve0, avg = agg_block_solver(agg, na)
    def M(x):
        rs = [x]
        for lev in lv[:-1]:
            rs.append(lev["P"].T @ rs[-1])
        y = None
        for k in range(len(lv) - 1, 0, -1):
            lev = lv[k]
            yk = np.einsum("sij,sj->si", lev["dinv"], rs[k].reshape(lev["nb"], 6)).ravel()
            if y is not None: yk = yk + lev["P"] @ y
            y = yk
        return solve0(x) + lv[0]["P"] @ y
    run("additive, aggregate blocks on the fine level (theta %.2f, %d aggregates, %.0f dof each)" % (theta, na, avg), M)
    run("aggregate blocks alone", solve0)
