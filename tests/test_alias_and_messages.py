"""`import gridpp` resolves to this implementation (guarded: an installed reference wins), and the message helpers of
include/gridpp.h:1394-1430 exist with the reference's output format (src/api/util.cpp:226-252)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_import_gridpp_is_this_implementation():
    import gridpp
    import gridpp_amd
    assert gridpp.implementation == "gridpp_amd"
    for name in ("Grid", "Points", "Point", "BarnesStructure", "CressmanStructure", "CrossValidation", "MultipleStructure", "optimal_interpolation",
                 "optimal_interpolation_full", "optimal_interpolation_ensi", "neighbourhood", "neighbourhood_quantile", "neighbourhood_quantile_fast",
                 "neighbourhood_brute_force", "get_neighbourhood_thresholds", "nearest", "bilinear", "Mean", "Median", "Geodetic", "Cartesian",
                 "set_omp_threads", "get_omp_threads", "set_debug_level", "get_debug_level", "version", "is_valid", "MV", "get_statistic"):
        assert getattr(gridpp, name) is getattr(gridpp_amd, name), name


def test_an_installed_reference_is_not_shadowed(tmp_path):
    """another `gridpp` on a different path entry wins (the alias only fills the gap when the reference is not installed)"""
    other = tmp_path / "site" / "gridpp"
    other.mkdir(parents=True)
    (other / "__init__.py").write_text("implementation = 'the installed reference'\n")
    code = "import sys; sys.path.insert(0, %r); sys.path.append(%r); import gridpp; print(gridpp.implementation)" % (ROOT, str(tmp_path / "site"))
    env = dict(os.environ)
    env.pop("GRIDPP_USE_AMD", None)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=str(tmp_path))
    assert out.stdout.strip() == "the installed reference", out.stdout + out.stderr
    env["GRIDPP_USE_AMD"] = "1"
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=str(tmp_path))
    assert out.stdout.strip() == "gridpp_amd", out.stdout + out.stderr


def test_debug_level_and_messages(capsys):
    import gridpp_amd as gridpp
    assert gridpp.get_debug_level() == 0
    gridpp.set_debug_level(3)
    assert gridpp.get_debug_level() == 3
    gridpp.set_debug_level(0)
    gridpp.warning("something")
    gridpp.debug("plain")
    gridpp.future_deprecation_warning("old", "new")
    gridpp.future_deprecation_warning("older")
    out = capsys.readouterr().out.splitlines()
    assert out == ["Warning: something", "plain", "Future deprecation warning: old will be deprecated, use new instead.",
                   "Future deprecation warning: older will be deprecated."]
    try:
        gridpp.error("bad")
        assert False
    except RuntimeError as e:
        assert str(e) == "bad"
    assert capsys.readouterr().out.strip() == "Error: bad"
    assert abs(gridpp.clock() - __import__("time").time()) < 5


def test_get_statistic_name_table():
    """include/gridpp.h:1410, src/api/gridpp.cpp:11-43: nine names; "variance" is not among them in the reference either"""
    import gridpp
    for name, st in (("mean", gridpp.Mean), ("min", gridpp.Min), ("max", gridpp.Max), ("median", gridpp.Median), ("quantile", gridpp.Quantile),
                     ("std", gridpp.Std), ("sum", gridpp.Sum), ("count", gridpp.Count), ("randomchoice", gridpp.RandomChoice)):
        assert gridpp.get_statistic(name) == st
    for name in ("variance", "Mean", "", "nonsense"):
        assert gridpp.get_statistic(name) == gridpp.Unknown
