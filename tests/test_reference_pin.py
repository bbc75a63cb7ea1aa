"""The pin to the real reference: tests/golden/ref_<name>.npz hold what MCMCDiagnosticTools.jl itself returns for the
inputs of the committed fixtures (written by julia/gen_reference_golden.jl on a machine with Julia; the build image
has none, so the files may be absent -- then the comparisons against real reference output skip and parity stays
"unpinned by reference output").  When present: the oracle must reproduce them (CPU), and so
must the HIP path (GPU).

The comparison code itself is exercised either way: `test_pin_comparison_covers_every_key` writes a SYNTHETIC
ref_<name>.npz for every fixture -- the fixture's own (oracle-minted) numbers under the keys and in the array shapes
the Julia generator writes -- runs the CPU comparison on it and asserts that every key of every such file was
compared.  That proves the route works the first time someone has Julia; it pins nothing."""
import glob
import os
import warnings

import numpy as np
import pytest

from conftest import GOLDEN_DIR, assert_close, load_golden

REFS = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "ref_*.npz")))
THREW = -12345.0
RTOL = 1e-9          # the contract is 1e-6 (BASELINE.json north_star); the restatements agree with each other to 1e-12
RTOL32 = 2e-4        # Float32: the reference computes in Float32, the oracle's comparator in Float64


def _input_of(ref_name, golden_dir=GOLDEN_DIR):
    """(fixture name, x) of ref_<name>: est_<base> / diag_<base> share the input of <base>"""
    name = ref_name[len("ref_"):]
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    if "x" in g:
        return name, g["x"]
    return name, np.load(os.path.join(golden_dir, name.split("_", 1)[1] + ".npz"))["x"]


def _threw(v):
    v = np.asarray(v)
    return v.size == 1 and float(v.ravel()[0]) == THREW


def generator_keys(fixture):
    """the keys julia/gen_reference_golden.jl writes for a fixture with these keys (reference_answers there)"""
    out = set()
    for k in fixture:
        p = k.split("|")
        if p[0] == "ess" and len(p) == 5:
            out |= {k, "rhat|" + k[4:]}
        elif (p[0] in ("essF", "rhatF") and len(p) == 5) or (p[0] == "nested" and len(p) == 3) or \
                (p[0] == "est" and len(p) == 5) or (p[0] == "err" and len(p) == 6 and p[1] == "est") or \
                (p[0] == "mcseq" and len(p) == 3) or (p[0] == "gelman" and len(p) == 3) or \
                k in ("mcse_mean", "bulk_tail_fft", "mcse_std", "bfmi", "stage_rank0", "stage_rank1"):
            out.add(k)
    if "stage_rank" in fixture:
        out |= {"stage_rank", "stage_z", "stage_fold", "stage_q05"}
    return out


class OracleBackend:
    """the CPU restatement (oracle/c_oracle.py) behind the calls the comparison makes"""
    z_tol = 1e-13

    def __init__(self):
        from oracle import c_oracle
        self.co = c_oracle
        self.errors = (c_oracle.OracleError,)

    def ess_rhat(self, x, kind, method, s, ml):
        return self.co.ess_rhat(x, kind=kind, autocov=method, split_chains=s, maxlag=ml)

    def bulk_tail_fft(self, x):
        return np.stack(self.co.bulk_tail(x, autocov="fft"))

    def nested(self, x, kind, s):
        inds = np.arange(x.shape[1], dtype=np.int32).reshape(x.shape[1] // 4, 4, order="F")
        return self.co.rhat_nested(x, inds, kind=kind, split_chains=s)

    def mcse_mean(self, x):
        return self.co.mcse_mean(x)

    def ess_est(self, x, est, p, method, s):
        return self.co.ess_estimator(x, est, prob=p, autocov=method, split_chains=s)

    def mcse_std(self, x):
        return self.co.mcse_std(x)

    def mcse_q(self, x, est, p):
        return self.co.mcse_quantile(x, est, prob=p)

    def gelman(self, x, alpha, which):
        return self.co.gelmandiag(x, alpha=alpha)[0 if which == "psrf" else 1]

    def bfmi(self, x):
        return self.co.bfmi(x.reshape(x.shape[0], -1, order="F"))

    def tiedrank(self, xp):
        return self.co.tiedrank(xp)                      # flat, column-major order

    def z(self, xp):
        return self.co.rank_normalize(xp)

    def fold(self, xp):
        return self.co.fold_around_median(xp)

    def q(self, xp, p):
        return self.co.quantile(xp, p)


class DeviceBackend:
    """the HIP path through the package (ctypes -> C-ABI)"""
    z_tol = 1e-11   # relative; the device's score table is good to ~1e-15 ABSOLUTE, scores near 0 are ~1e-3

    def __init__(self):
        import torch
        import __graft_entry__ as ge
        self.torch = torch
        self.pkg = ge.load_package()
        p = self.pkg
        self.methods = {"direct": p.AutocovMethod(), "fft": p.FFTAutocovMethod(), "bda": p.BDAAutocovMethod()}
        self.errors = (p.ArgumentError, p.DomainError, p.DimensionMismatch)

    def dev(self, x):
        return self.torch.from_numpy(np.ascontiguousarray(np.transpose(x, (2, 1, 0)))).cuda().permute(2, 1, 0)

    @staticmethod
    def host(t):
        return t.cpu().numpy() if hasattr(t, "cpu") else np.asarray(t)

    def ess_rhat(self, x, kind, method, s, ml):
        r = self.pkg.ess_rhat(self.dev(x), kind=kind, autocov_method=self.methods[method], split_chains=s, maxlag=ml)
        return self.host(r.ess), self.host(r.rhat)

    def bulk_tail_fft(self, x):
        return np.stack([self.host(o) for o in self.pkg.bulk_tail(self.dev(x), autocov_method=self.methods["fft"])])

    def nested(self, x, kind, s):
        ids = np.repeat(np.arange(1, 5), x.shape[1] // 4)
        return self.host(self.pkg.rhat_nested(self.dev(x), ids, kind=kind, split_chains=s))

    def mcse_mean(self, x):
        return self.host(self.pkg.mcse(self.dev(x)))

    def _kind(self, est, p):
        return {"median": np.median, "std": np.std, "mad": "mad"}.get(est) or self.pkg.Quantile(p)

    def ess_est(self, x, est, p, method, s):
        return self.host(self.pkg.ess(self.dev(x), kind=self._kind(est, p), autocov_method=self.methods[method], split_chains=s))

    def mcse_std(self, x):
        return self.host(self.pkg.mcse(self.dev(x), kind=np.std))

    def mcse_q(self, x, est, p):
        return self.host(self.pkg.mcse(self.dev(x), kind=self._kind(est, p)))

    def gelman(self, x, alpha, which):
        return self.host(getattr(self.pkg.gelmandiag(self.dev(x), alpha=alpha), which))

    def bfmi(self, x):
        e = x.reshape(x.shape[0], -1, order="F")
        return self.host(self.pkg.bfmi(self.dev(e[:, :, None])[:, :, 0]))

    def _stage(self, xp, fold=False):
        ranks, z = self.pkg.rank_normalize(self.dev(xp[:, :, None]), fold=fold)
        return self.host(ranks)[:, :, 0], self.host(z)[:, :, 0]

    def tiedrank(self, xp):
        return self._stage(xp)[0].ravel(order="F")

    def z(self, xp):
        return self._stage(xp)[1]

    fold = None   # (the folded values themselves are not an output of the device path; its ranks are checked elsewhere)
    q = None


def compare_with_reference(ref, x, be, what):
    """every key of a ref_<name>.npz against backend `be`; returns the set of keys that were compared"""
    f32 = x.dtype == np.float32
    tol = RTOL32 if f32 else RTOL
    done = set()

    def expect(key, call, want, tol_=None):
        """the call must throw where the reference threw, and agree with it where it did not"""
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if _threw(want):
                with pytest.raises(be.errors):
                    call()
            else:
                got = call()
                assert_close(np.asarray(got, dtype=np.float64).ravel(order="F"), np.asarray(want, dtype=np.float64).ravel(order="F"),
                             tol if tol_ is None else tol_, f"{what} {key}")
        done.add(key)

    for key, want in ref.items():
        p = key.split("|")
        if p[0] in ("ess", "essF") and len(p) == 5:
            rk = ("rhat|" if p[0] == "ess" else "rhatF|") + key.split("|", 1)[1]
            kw = (x, p[1], p[2], int(p[3]), int(p[4]))
            expect(key, lambda: be.ess_rhat(*kw)[0], want)
            if rk in ref:
                expect(rk, lambda: be.ess_rhat(*kw)[1], ref[rk])
        elif p[0] in ("rhat", "rhatF") and len(p) == 5:
            if ("ess|" if p[0] == "rhat" else "essF|") + key.split("|", 1)[1] not in ref:  # (else: compared with its ess twin)
                expect(key, lambda: be.ess_rhat(x, p[1], p[2], int(p[3]), int(p[4]))[1], want)
        elif key == "bulk_tail_fft":
            expect(key, lambda: be.bulk_tail_fft(x), np.asarray(want).reshape(4, -1) if not _threw(want) else want)
        elif p[0] == "nested" and len(p) == 3:
            expect(key, lambda: be.nested(x, p[1], int(p[2])), want)
        elif key == "mcse_mean":
            expect(key, lambda: be.mcse_mean(x), want)
        elif p[0] == "est" and len(p) == 5:
            expect(key, lambda: be.ess_est(x, p[1], float(p[2]), p[3], int(p[4])), want)
        elif p[0] == "err" and len(p) == 6 and p[1] == "est":
            # the fixture expects a throw here (Statistics.quantile on NaN data); the reference's verdict decides
            expect(key, lambda: be.ess_est(x, p[2], float(p[3]), p[4], int(p[5])), want)
        elif key == "mcse_std":
            expect(key, lambda: be.mcse_std(x), want)
        elif p[0] == "mcseq" and len(p) == 3:
            expect(key, lambda: be.mcse_q(x, p[1], float(p[2])), want)
        elif p[0] == "gelman" and len(p) == 3:
            expect(key, lambda: be.gelman(x, float(p[1]), p[2]), want, 2e-6 if f32 else RTOL)
        elif key == "bfmi":
            expect(key, lambda: be.bfmi(x), want, 2e-5 if f32 else RTOL)
        elif key in ("stage_rank", "stage_rank0", "stage_rank1"):
            xp = x[:, :, 1 if key == "stage_rank1" else 0]
            # the generator writes the ranks in the shape of the slice, (draws, chains); the oracle returns the pooled vector
            assert np.array_equal(np.asarray(be.tiedrank(xp)).ravel(order="F"), np.asarray(want, dtype=np.float64).ravel(order="F")), f"{what} {key}"
            done.add(key)
        elif key == "stage_z":
            assert_close(np.asarray(be.z(x[:, :, 0])).ravel(order="F"), np.asarray(want).ravel(order="F"), 1e-6 if f32 else be.z_tol, f"{what} {key}")
            done.add(key)
        elif key == "stage_fold":
            if be.fold is not None:
                assert np.array_equal(np.asarray(be.fold(x[:, :, 0])).ravel(order="F"), np.asarray(want).ravel(order="F")), f"{what} {key}"
            done.add(key)
        elif key == "stage_q05":
            if be.q is not None:
                assert np.array_equal([be.q(x[:, :, 0], 0.05), be.q(x[:, :, 0], 0.95)], np.asarray(want).ravel()), f"{what} {key}"
            done.add(key)
    return done


def synthetic_reference(name, golden_dir=GOLDEN_DIR):
    """what gen_reference_golden.jl would write for fixture `name` IF the reference returned the fixture's own numbers:
    same keys, and the array shapes Julia gives them (ranks and stage arrays as (draws, chains) matrices, every result
    vector Float64, `err|` keys as the throw sentinel)"""
    g = dict(np.load(os.path.join(golden_dir, name + ".npz")))
    _, x = _input_of("ref_" + name, golden_dir)
    D, C = x.shape[0], x.shape[1]
    out = {}
    for k in generator_keys(g):
        if k not in g and k.startswith("stage_"):   # (written by the generator wherever stage_rank is; not every fixture holds them)
            from oracle import c_oracle as co
            xp = x[:, :, 0]
            g[k] = {"stage_z": lambda: co.rank_normalize(xp), "stage_fold": lambda: co.fold_around_median(xp),
                    "stage_q05": lambda: np.array([co.quantile(xp, 0.05), co.quantile(xp, 0.95)])}[k]()
        if k.startswith("err|"):
            out[k] = np.array([THREW])
        elif k in ("stage_rank", "stage_rank0", "stage_rank1"):
            out[k] = np.asarray(g[k], dtype=np.float64).reshape(D, C, order="F")
        elif k in ("stage_z", "stage_fold"):
            out[k] = np.asarray(g[k]).reshape(D, C, order="F")
        else:
            out[k] = np.asarray(g[k], dtype=np.float64)
    return out


def fixture_names():
    names = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    return [n for n in names if not n.startswith("ref_") and n != "diag_bfmi_reference"]


def test_generator_is_committed():
    """the route to a pinned oracle exists even where the reference output does not"""
    root = os.path.dirname(GOLDEN_DIR)
    assert os.path.exists(os.path.join(os.path.dirname(root), "julia", "gen_reference_golden.jl"))


@pytest.mark.parametrize("name", fixture_names())
def test_pin_comparison_covers_every_key(name, tmp_path):
    """the comparison the real pin will run, executed on a synthetic ref_<name>.npz: every key the generator writes for
    every fixture is compared (none silently ignored), in the shapes the generator writes them"""
    ref = synthetic_reference(name)
    assert ref, name
    np.savez(tmp_path / ("ref_" + name + ".npz"), **ref)          # through a file, like the real thing
    loaded = dict(np.load(tmp_path / ("ref_" + name + ".npz")))
    _, x = _input_of("ref_" + name)
    done = compare_with_reference(loaded, x, OracleBackend(), name)
    assert done == set(loaded), sorted(set(loaded) - done)
    fixture = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    not_pinned = set(fixture) - generator_keys(fixture) - {"x", "inds"}
    assert all(k.startswith(("stage_acov_", "rhat|", "rhatF|")) for k in not_pinned), sorted(not_pinned)


@pytest.mark.skipif(not REFS, reason="no tests/golden/ref_*.npz: run julia/gen_reference_golden.jl where Julia is available")
@pytest.mark.parametrize("ref", REFS)
def test_oracle_reproduces_the_reference(ref):
    name, x = _input_of(ref)
    r = load_golden(ref)
    done = compare_with_reference(r, x, OracleBackend(), ref)
    assert done == set(r), sorted(set(r) - done)


@pytest.mark.gpu
@pytest.mark.skipif(not REFS, reason="no tests/golden/ref_*.npz: run julia/gen_reference_golden.jl where Julia is available")
@pytest.mark.parametrize("ref", REFS)
def test_device_reproduces_the_reference(ref):
    name, x = _input_of(ref)
    r = load_golden(ref)
    done = compare_with_reference(r, x, DeviceBackend(), ref)
    assert done == set(r), sorted(set(r) - done)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cfg1_randn_1000x4x10", "est_ties_200x4x6", "diag_f32_300x4x6", "special_50x2x3", "nested_200x16x4", "f32_300x4x6"])
def test_pin_comparison_runs_on_the_device_path(name):
    """the device half of the comparison, on synthetic references of a few fixtures (so that it, too, has run before
    real reference output exists)"""
    ref = synthetic_reference(name)
    _, x = _input_of("ref_" + name)
    done = compare_with_reference(ref, x, DeviceBackend(), name)
    assert done == set(ref), sorted(set(ref) - done)
