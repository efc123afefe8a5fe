"""tests/golden/make_golden_from_reference.py: the plumbing of the reference-side generator, exercised WITHOUT the reference
(gpflow 1.1.1 / TF 1.8 do not exist here): the dry run puts the oracle in the reference's place, and a stub with a wrong number must
be reported.  This proves the comparison / writing code, not parity — DESIGN.md section 3 keeps "parity unpinned"."""
import numpy as np

from tests.golden import cases
from tests.golden import make_golden_from_reference as G


def test_without_the_reference_environment_it_says_so_and_changes_nothing(capsys):
    assert G.main(["--reference", "/nonexistent/reference", "--check-only"]) == 0
    out = capsys.readouterr().out
    assert "cannot run here" in out and "UNPINNED" in out


def test_dry_run_writes_complete_fixtures_elsewhere(tmp_path, capsys):
    names = ["two_layer_1d", "multiclass"]
    assert G.main(["--dry-run", "--out-dir", str(tmp_path), "--cases", ",".join(names)]) == 0
    assert "largest relative deviation vs committed fixtures: 0.000e+00" in capsys.readouterr().out
    for n in names:
        new = np.load(tmp_path / f"golden_{n}.npz")
        old = np.load(f"{G.HERE}/golden_{n}.npz")
        assert set(old.files) <= set(new.files) and str(new["source"]).startswith("oracle")
        for k in old.files:
            assert k == "source" or np.array_equal(old[k], new[k]), k


def test_dry_run_refuses_to_overwrite_the_committed_fixtures():
    assert G.main(["--dry-run", "--cases", "two_layer_1d"]) == 2


def test_a_deviating_or_incomplete_generator_is_reported(tmp_path, capsys):
    out = G.oracle_stand_in("two_layer_1d")
    out["elbo"] = out["elbo"] * (1 + 1e-5)
    assert G.compare_and_write("two_layer_1d", out, str(tmp_path), check_only=True) > 5e-6
    assert "two_layer_1d:elbo deviates" in capsys.readouterr().out
    del out["x.preddens"]
    assert G.compare_and_write("two_layer_1d", out, str(tmp_path), check_only=True) == float("inf")
    assert not list(tmp_path.iterdir())          # check_only: nothing written


def test_reference_side_covers_every_key_of_the_oracle_side():
    """the generator's source names every extras key the fixtures hold (a key added to extras.py must be added there too)"""
    src = open(G.__file__).read()
    old = np.load(f"{G.HERE}/golden_svgp_matern52.npz")
    for k in old.files:
        if k.startswith("x."):
            assert f'"{k}"' in src or f'"{k[:-5]}"' in src or f'"{k[:-6]}"' in src, k
    # (golden_full_*.npz are the full-size oracle fixtures of tests/golden/full_cases.py: their own generator and test)
    assert set(cases.CASES) == {f[len("golden_"):-4] for f in __import__("os").listdir(G.HERE)
                                if f.endswith(".npz") and not f.startswith("golden_full_")}
