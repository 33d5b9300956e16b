"""The correlation tables (ba_correlations_compute / ba_print_correlations restating the COMPUTECORRELATION blocks,
src/main.cpp:5592-6495, with extractAPCorrelation :837 and calcStatistics :851) against a numpy computation from the
full covariance matrix.  The reference has the block compiled off (COMPUTECORRELATION 0, :47) and holds no output of
it: the layout of the text follows its std::cout statements, the numbers are checked against numpy."""
import numpy as np

from bundleadjustment_amd import host
from oracle import py_oracle as po
from tests.test_oracle_covariance import numpy_covariance


def numpy_tables(net, camera):
    S, P = net.n_sta, net.n_pts
    Cx, cam_col = numpy_covariance(net)
    sd = np.sqrt(np.diag(Cx))
    out = dict(eop_iop=np.full((4, 6, 3), np.nan), eop_ap=np.full((4, 6, 16), np.nan), xyz_iop=np.full((4, 3, 3), np.nan),
               xyz_ap=np.full((4, 3, 16), np.nan), iop_ap=np.full((3, 16), np.nan), ap_ap=np.full((16, 16), np.nan))
    for k in ("eop_iop", "eop_ap", "xyz_iop", "xyz_ap"):
        out[k][3] = 0.0  # calcStatistics leaves min at zero for an empty list and the reference does not overwrite it
    for q in range(19):
        col = cam_col.get((camera, q))
        if col is None:
            continue
        for m in range(6):
            idx = 6 * np.arange(S) + m
            c = np.abs(Cx[col, idx] / (sd[idx] * sd[col]))
            t = out["eop_iop"][:, m, q] if q < 3 else out["eop_ap"][:, m, q - 3]
            t[:] = [c.max(), np.median(c), c.mean(), c.min()]
        for m in range(3):
            idx = 6 * S + 3 * np.arange(P) + m
            c = np.abs(Cx[col, idx] / (sd[idx] * sd[col]))
            t = out["xyz_iop"][:, m, q] if q < 3 else out["xyz_ap"][:, m, q - 3]
            t[:] = [c.max(), np.median(c), c.mean(), c.min()]
        for q2 in range(3, 19):
            col2 = cam_col.get((camera, q2))
            if col2 is None:
                continue
            r = Cx[col, col2] / (sd[col] * sd[col2])
            if q < 3:
                out["iop_ap"][q, q2 - 3] = r
            else:
                out["ap_ap"][q - 3, q2 - 3] = r
    return out


def test_correlation_tables_match_numpy(tmp_path):
    net = host.synth(n_cam=2, n_sta=8, n_pts=60, rays_per_point=5, ap_free_mask=host.AP_K1K2P1P2, control_every=4,
                     outlier_fraction=0.1, seed=3).net
    var, rows = po.covariance(net), po.covariance_rows(net)
    for camera in (0, 1):
        path = tmp_path / f"corr{camera}.txt"
        got = host.correlations(var, rows, camera=camera, path=str(path), text=True)
        want = numpy_tables(net, camera)
        for k, w in want.items():
            assert got[k].shape == w.shape, k
            assert np.array_equal(np.isnan(got[k]), np.isnan(w)), k
            assert np.allclose(got[k], w, rtol=0, atol=1e-8, equal_nan=True), k
        lines = path.read_text().splitlines()
        # the reference's layout: AP-AP matrix (header, two statistics lines, 16 rows), then the EOP-IOP banner
        assert lines[0] == "Correlation Matrix AP-AP..." and lines[1].startswith("   Mean(fabs): ") and lines[2].startswith("   Median(fabs): ")
        assert lines[19] == "============================= Correlation EOP-IOP ============================="
        assert lines[20] == "Correlation EOP-IOP Max(fabs)" and lines[23] == "       xp\typ\tc"
        assert lines[24].startswith("omega: ") and lines[29].startswith("Zo   : ")
        assert [float(x) for x in lines[24].split()[1:]] == [round(float(x), 2) for x in want["eop_iop"][0, 0]]
        heads = [l for l in lines if l.startswith("Correlation ")]
        assert heads == ["Correlation Matrix AP-AP..."] + [f"Correlation {a} {b}(fabs)" for a in ("EOP-IOP", "EOP-AP", "XYZ-IOP")
                                                           for b in ("Max", "Median", "Mean", "Min")] + \
            ["Correlation XYZ-AP Max(fabs)", "Correlation IOP-AP"]
        # the off-diagonal statistics of the AP-AP matrix (:5601)
        off = want["ap_ap"][~np.eye(16, dtype=bool)]
        off = np.abs(off[~np.isnan(off)])
        assert lines[1] == "   Mean(fabs): %.2f +/- %.2f" % (off.mean(), off.std())
        assert lines[2] == "   Median(fabs): %.2f (%.2f to %.2f)" % (np.median(off), off.min(), off.max())


def test_correlations_without_point_rows_and_with_a_fixed_camera():
    net = host.synth(n_cam=1, n_sta=9, n_pts=80, rays_per_point=5, ap_free_mask=0, iop_free_mask=0, seed=7, control_every=8).net
    var = po.covariance(net)
    par, ce, cx, cc = po.covariance_rows(net)
    assert len(par) == 0
    got = host.correlations(var, (par, ce, None, cc))
    for k in ("eop_iop", "eop_ap", "xyz_iop", "xyz_ap"):
        assert np.all(np.isnan(got[k][:3])) and np.all(got[k][3] == 0.0)
    assert np.all(np.isnan(got["iop_ap"])) and np.all(np.isnan(got["ap_ap"]))
