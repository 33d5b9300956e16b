"""The command-line program (host/ba_main.cpp: reader -> ba_create -> ba_solve -> ba_covariance -> reports and result
files) run as the reference's bundleAdjustment binary would be, against the same steps taken through the Python
bindings."""
import os
import subprocess

import numpy as np
import pytest

from bundleadjustment_amd import build, capi, host, solver

pytestmark = pytest.mark.gpu


def run_cli(args, cwd):
    exe = build.MAIN_BIN
    assert os.path.exists(exe), "ba_main is not built: python -m bundleadjustment_amd.build"
    return subprocess.run([exe] + args, cwd=cwd, capture_output=True, text=True, timeout=240)


def table(path):
    return np.loadtxt(path, ndmin=2)


@pytest.fixture()
def files(tmp_path):
    hn = host.synth(n_cam=2, n_sta=14, n_pts=160, rays_per_point=6, seed=21, ap_free_mask=host.AP_K1K2P1P2)
    hn.write_inputs(str(tmp_path), "net")
    return tmp_path, [str(tmp_path / ("net." + e)) for e in ("pho", "eop", "iop", "xyz")]


def test_cli_matches_the_bindings(files):
    tmp, paths = files
    out = tmp / "cli"
    out.mkdir()
    r = run_cli(paths + ["--free-ap", hex(host.AP_K1K2P1P2), "--correlations", "--out", str(out)], str(tmp))
    assert r.returncode == 0, r.stderr
    assert "Termination CONVERGENCE" in r.stdout

    # the same through Python, from the same files
    hn = host.read_network(*paths)
    hn.set_free(host.AP_K1K2P1P2, 7)
    ref = tmp / "py"
    ref.mkdir()
    with solver.Problem(hn.net) as P:
        eop, xyz, iop, ap, sm, rows = P.solve(solver.default_options())
        var = P.covariance()
        cam_rows = P.covariance_camera_rows()
        _, res = P.evaluate(eop, xyz, iop, ap)
    # the correlation tables of the last camera (what the reference's matrices hold), as printed by the tool
    host.correlations(var, cam_rows, camera=hn.net.n_cam - 1, path=str(tmp / "corr.txt"), text=True)
    block = (tmp / "corr.txt").read_text()
    assert block.startswith("Correlation Matrix AP-AP...") and block in r.stdout
    hn.set_parameters(eop, xyz, iop, ap)
    hn.set_variances(*var)
    hn.write_angles(str(ref))
    hn.write_results(str(ref), res)
    names = ("EOP.jck", "temp.eop", "iop.jck", "temp.iop", "ap.jck", "XYZ.jck", "image.jck", "temp.pho", "angles.jck")
    for name in names:
        a, b = table(out / name), table(ref / name)
        assert a.shape == b.shape, name
        # two solves of the same problem are bitwise reproducible, so the printed files are identical
        assert np.array_equal(a, b), name
    # the iteration table and the summary line carry the solve
    assert ("Final cost %.6e" % sm["final_cost"]) in r.stdout
    assert len([ln for ln in r.stdout.splitlines() if ln[:4].strip().isdigit()]) >= sm["num_iterations"]


def test_cli_reports_the_empty_box_of_a_centred_sensor_format(files):
    tmp, paths = files
    r = run_cli(paths + ["--model", "4", "--iop-bounds", "--out", str(tmp)], str(tmp))
    # the generator's format columns are +-2000: the reference's yp interval [-yMax, yMin] is empty
    assert r.returncode == 1 and "empty interval" in r.stderr


def test_cli_synthetic_network_with_a_box(tmp_path):
    # image rows counted from yMin = 0, as the reference's formula expects: rewrite the format columns of the .iop file
    hn = host.synth(n_cam=1, n_sta=12, n_pts=150, rays_per_point=5, seed=8, ap_free_mask=host.AP_K1K2P1P2)
    hn.write_inputs(str(tmp_path), "net")
    iop_path = tmp_path / "net.iop"
    rows = [ln.split() for ln in iop_path.read_text().splitlines() if ln.strip()]
    for t in rows:
        t[2:6] = ["-2000", "10", "2000", "2000"]  # xp in [-2000, 2000], yp in [-2000, 10]
    iop_path.write_text("".join(" ".join(t) + "\n" for t in rows))
    paths = [str(tmp_path / ("net." + e)) for e in ("pho", "eop", "iop", "xyz")]
    a = run_cli(paths + ["--free-ap", hex(host.AP_K1K2P1P2), "--out", str(tmp_path)], str(tmp_path))
    assert a.returncode == 0, a.stderr
    free = table(tmp_path / "iop.jck")
    b = run_cli(paths + ["--free-ap", hex(host.AP_K1K2P1P2), "--iop-bounds", "--out", str(tmp_path)], str(tmp_path))
    assert b.returncode == 0, b.stderr
    boxed = table(tmp_path / "iop.jck")
    # the box is wide: same adjustment, by the bounds-constrained route
    assert np.allclose(free[:, 1:4], boxed[:, 1:4], rtol=0, atol=1e-6)
