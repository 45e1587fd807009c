"""Host-side logic that needs no GPU: the hyper-parameter optimiser state machine, the Logexp
transform, the LCBSC argument handling and random-number consumption.

The SCG restatements in elfi_amd/hyperopt.py (product) and oracle/gp_hyper_oracle.py (checker)
are written independently; on analytic objectives they must walk the same trajectory.
"""
import pickle

import numpy as np
import pytest

import gp_hyper_oracle as HO
import gp_oracle as G
from elfi_amd import hyperopt as H


def _rosen(x):
    return float(np.sum(100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2))


def _rosen_grad(x):
    g = np.zeros_like(x)
    g[:-1] = -400 * x[:-1] * (x[1:] - x[:-1] ** 2) - 2 * (1 - x[:-1])
    g[1:] += 200 * (x[1:] - x[:-1] ** 2)
    return g


@pytest.mark.parametrize('dim,iters', [(2, 50), (4, 50), (4, 400)])
def test_scg_product_and_oracle_walk_the_same_path(dim, iters):
    x0 = np.linspace(-1.2, 1.0, dim)
    xa, fa, nfev, sa = H.scg(_rosen, _rosen_grad, x0, maxiters=iters)
    xb, fb, sb = HO.scaled_conjugate_gradient(_rosen, _rosen_grad, x0, iterations=iters)
    assert sa == sb
    assert len(fa) == len(fb)
    np.testing.assert_allclose(fa, fb, rtol=1e-12, atol=0)
    np.testing.assert_allclose(xa, xb, rtol=1e-12, atol=1e-14)
    assert fa[-1] < fa[0]
    assert all(b <= a + 1e-12 for a, b in zip(fa, fa[1:])), 'SCG never accepts an uphill step'


def test_scg_converges_on_a_quadratic():
    A = np.diag([1.0, 10.0, 100.0])
    b = np.array([1.0, -2.0, 3.0])
    x, flog, nfev, status = H.scg(lambda x: 0.5 * x @ A @ x - b @ x, lambda x: A @ x - b, np.zeros(3),
                                  maxiters=200)
    np.testing.assert_allclose(x, np.linalg.solve(A, b), rtol=5e-3, atol=1e-5)  # stops on |df| < 1e-6
    assert status.startswith('converged')


def test_logexp_transform_roundtrip_and_gradfactor():
    theta = np.array([1e-6, 0.01, 0.3, 1.0, 5.0, 35.9, 36.1, 100.0])
    phi = H.logexp_inv(theta)
    np.testing.assert_allclose(H.logexp(phi), theta, rtol=1e-12)
    np.testing.assert_allclose(HO.softplus(phi), theta, rtol=1e-12)
    np.testing.assert_allclose(HO.softplus_inv(theta), phi, rtol=1e-12, atol=1e-12)
    eps = 1e-6
    fd = (H.logexp(phi + eps) - H.logexp(phi - eps)) / (2 * eps)
    np.testing.assert_allclose(H.logexp_gradfactor(theta), fd, rtol=1e-6)


def test_oracle_map_objective_gradient_is_consistent():
    """Finite differences on the CPU oracle's objective (pins the formula the GPU is tested against)."""
    X, y, bounds = G.synthetic_gp_problem(60, 3, seed=4)
    pri = G.default_priors(bounds, y)
    obj = HO.MapObjective(X, y, pri)
    phi = HO.softplus_inv(np.array([0.7, 1.3, 0.4, 0.2]))
    g = obj.gradient(phi)
    for i in range(4):
        e = np.zeros(4)
        e[i] = 1e-6
        fd = (obj.value(phi + e) - obj.value(phi - e)) / 2e-6
        assert abs(fd - g[i]) <= 1e-5 * (1 + abs(g[i])), (i, fd, g[i])


def test_oracle_optimize_improves_the_objective():
    X, y, bounds = G.synthetic_gp_problem(80, 2, seed=2)
    pri = G.default_priors(bounds, y)
    h0 = G.initial_hyper(y)
    h, info = HO.optimize(X, y, h0, pri, max_iters=50)
    assert info['objective'][-1] < info['objective'][0] - 1.0
    assert all(v > 0 for v in h.values())


# ---- LCBSC host logic --------------------------------------------------------------------
class _Model:
    parameter_names = ['a', 'b']
    input_dim = 2
    bounds = [(-2, 2), (-1, 1)]
    n_evidence = 0
    _handle = None


def test_lcbsc_arguments_and_beta():
    from elfi_amd import HipLCBSC
    acq = HipLCBSC(_Model(), exploration_rate=10, seed=0)
    # acquisition.py:256-260: beta_t = 2 log((t+1)^(2d+2) pi^2 / (3 delta)), delta = 1/exploration_rate
    for t in (0, 3, 99):
        assert acq._beta(t) == pytest.approx(2 * np.log((t + 1) ** 6 * np.pi ** 2 / (3 * 0.1)))
        assert acq._beta(t) == pytest.approx(G.lcb_beta(t, 2, 10.))
    assert HipLCBSC(_Model(), delta=0.25).exploration_rate == 4
    assert HipLCBSC(_Model(), noise_var={'b': 0.2, 'a': 0.1}).noise_var == [0.1, 0.2]
    assert HipLCBSC(_Model(), noise_var=0.3).noise_var == 0.3
    with pytest.raises(ValueError):
        HipLCBSC(_Model(), noise_var=-1.0)
    with pytest.raises(ValueError):
        HipLCBSC(_Model(), noise_var={'a': 0.1})
    with pytest.raises(ValueError):
        HipLCBSC(_Model(), noise_var={'a': 0.1, 'b': -0.2})
    with pytest.raises(ValueError):
        HipLCBSC(_Model(), noise_var='x')
    # constraints / additive_cost are accepted as the reference accepts them (acquisition.py:24-73); what they do is
    # tested on the GPU (tests/test_acquisition_gpu.py)
    cons = [{'type': 'ineq', 'fun': lambda x: x[0]}]
    assert HipLCBSC(_Model(), constraints=cons).constraints is cons


def test_lcbsc_random_number_consumption_matches_the_reference_recipe():
    """utils.py:72-79 draws the start points dimension by dimension from random_state.uniform;
    acquisition.py:174-191 then draws the jitter dimension by dimension from truncnorm."""
    import scipy.stats as ss
    from elfi_amd import HipLCBSC
    acq = HipLCBSC(_Model(), n_inits=7, noise_var=0.1, seed=123)
    sp = acq._start_points()
    rs = np.random.RandomState(123)
    exp = np.empty((7, 2))
    for i, b in enumerate(_Model.bounds):
        exp[:, i] = rs.uniform(*b, 7)
    assert np.array_equal(sp, exp)
    x = np.tile(sp[0], (3, 1))
    got = acq._add_noise(x.copy())
    for i, (lo, hi) in enumerate(_Model.bounds):
        std = np.sqrt(0.1)
        e = ss.truncnorm.rvs((lo - x[:, i]) / std, (hi - x[:, i]) / std, loc=x[:, i], scale=std, size=3,
                             random_state=rs)
        assert np.array_equal(got[:, i], e)
    assert np.all(got[:, 0] >= -2) and np.all(got[:, 0] <= 2) and np.all(np.abs(got[:, 1]) <= 1)


def test_lcbsc_without_evidence_returns_a_start_point():
    from elfi_amd import HipLCBSC
    acq = HipLCBSC(_Model(), seed=5)
    x = acq.acquire(4, t=0)
    assert x.shape == (4, 2) and np.all(x == x[0])


def test_hip_objects_pickle_without_device_state():
    """ELFI pickles operations and models (clients/multiprocessing.py:50, elfi_model.py:401-438)."""
    import elfi_amd
    op = elfi_amd.HipDiscrepancy('minkowski', p=3, w=[1., 2.])
    op2 = pickle.loads(pickle.dumps(op))
    assert op2.dist.metric == 'minkowski' and op2.dist.p == 3 and np.array_equal(op2.dist.w, [1., 2.])
    gp = elfi_amd.HipGPRegression(['a', 'b'], bounds={'a': (-2, 2), 'b': (-1, 1)})
    gp2 = pickle.loads(pickle.dumps(gp))
    assert gp2.bounds == [(-2, 2), (-1, 1)] and gp2.n_evidence == 0
    mu, var = gp2.predict(np.zeros((3, 2)))
    assert np.array_equal(mu, np.zeros((3, 1))) and np.array_equal(var, np.ones((3, 1)))   # :114-115
    st = elfi_amd.AdaptiveDistanceState()
    st2 = pickle.loads(pickle.dumps(st))
    assert st2.state['w'] == [None]


def test_loop_timers_split_update_search_and_acquire():
    """benchlib.loop_timing.instrument: wraps exactly the three calls the reference's BOLFI loop makes into the device
    objects (bolfi.py:219,247: target_model.update(..., optimize) and acquisition_method.acquire) and takes itself out again."""
    import time
    from benchlib.loop_timing import instrument

    class _GP:
        n_evidence = 0
        _opt_info = None

        def update(self, x, y, optimize=False):
            time.sleep(0.002)
            self.n_evidence += 1
            if optimize:
                self.optimize()

        def optimize(self):
            time.sleep(0.004)
            self._opt_info = dict(n_fits=7, status='ok')

    class _Acq:
        def acquire(self, n, t=None):
            time.sleep(0.001)
            return np.zeros((n, 1))

    gp, acq = _GP(), _Acq()
    T = instrument(gp, acq)
    for i in range(6):
        gp.update(None, None, optimize=(i % 3 == 2))
        acq.acquire(1, t=i)
    T.restore()
    assert T.updates == 6 and T.acquires == 6 and len(T.searches) == 2
    assert T.searches[0][0] == 3 and T.searches[0][2] == 7
    # (lower bounds are the sleeps; the upper bounds only have to tell the three timers apart on a loaded machine)
    assert 0.010 <= T.fit_s < 0.2 and 0.007 <= T.search_s < 0.2 and 0.005 <= T.acquire_s < 0.2
    assert 'update' not in gp.__dict__ and 'acquire' not in acq.__dict__
    s = T.summary(0.05, 6)
    assert abs(sum(s['share'].values()) - 1.0) < 1e-9 and s['rebuilds_in_searches'] == 14


def test_truncnorm_draw_is_the_public_call_bit_for_bit():
    """lcb_acquisition.truncnorm_draw (the jitter of acquire(), acquisition.py:174-191): same values and same generator
    state as ss.truncnorm.rvs, through the direct form and through the fall-backs."""
    import scipy.stats as ss
    import elfi_amd.lcb_acquisition as L
    rs1, rs2 = np.random.RandomState(3), np.random.RandomState(3)
    std = np.sqrt(0.1)
    for k in range(300):
        lo, hi = (0.0, 2.0) if k % 3 else (-1.0, 0.5)
        c = np.array([rs1.uniform(lo, hi)])
        assert rs2.uniform(lo, hi) == c[0]
        a, b = (lo - c) / std, (hi - c) / std
        x1 = ss.truncnorm.rvs(a, b, loc=c, scale=std, size=1, random_state=rs1)
        x2 = L.truncnorm_draw(a, b, c, std, 1, rs2)
        assert np.array_equal(x1, x2)
    assert L._TRUNCNORM_DIRECT in (True, False)
    # batches of several points take the public call
    c = np.array([0.3, 1.1, 1.9])
    x1 = ss.truncnorm.rvs((0 - c) / std, (2 - c) / std, loc=c, scale=std, size=3, random_state=rs1)
    x2 = L.truncnorm_draw((0 - c) / std, (2 - c) / std, c, std, 3, rs2)
    assert np.array_equal(x1, x2)
    assert np.array_equal(rs1.get_state()[1], rs2.get_state()[1]) and rs1.get_state()[2] == rs2.get_state()[2]


def test_lean_truncnorm_quantile_is_scipys_bit_for_bit():
    """The restatement of truncnorm_gen._ppf that acquire()'s jitter uses after its run-time check: every branch it
    accepts, against SciPy's own function, bit for bit (declines intervals on one side of 0)."""
    import scipy.stats as ss
    import elfi_amd.lcb_acquisition as L
    rs = np.random.RandomState(0)
    n_checked = 0
    for k in range(20000):
        q = rs.uniform(size=1)
        if k % 7 == 0:
            q = np.array([rs.choice([1e-300, 1e-17, 0.5, 1 - 1e-16, 1e-8])])
        lo, hi = sorted(rs.uniform(-40, 40, 2))
        if k % 3 == 0:
            lo, hi = -abs(lo) - 1e-3, abs(hi) + 1e-3
        if k % 11 == 0:
            lo, hi = 0.0, abs(hi) + 0.1
        if k % 13 == 0:
            lo, hi = -abs(lo) - 0.1, 0.0
        a, b = np.array([lo]), np.array([hi])
        x = L._truncnorm_ppf_lean(q, a, b)
        if x is None:
            assert not (lo <= 0.0 < hi)
            continue
        ref = ss.truncnorm._ppf(q, a, b)
        assert np.array_equal(ref, x) or (np.isnan(ref).all() and np.isnan(x).all()), (q, a, b, ref, x)
        n_checked += 1
    assert n_checked > 5000


def test_kept_registries_and_pinned_pool_plumbing():
    """elfi_amd._lib: the registries that let a distance / simulator result be recognised by identity later (weak
    references: an entry dies with its array, an id reused by another array does not match), views aliasing an entry,
    and the page-locked pool's fall-back to ordinary arrays where pinning is not possible (no GPU here)."""
    import gc
    from elfi_amd import _lib

    class FakeCtx:
        def __init__(self):
            self.epoch = 0

        def kept_epoch(self):
            return self.epoch

        def call(self, name, ep, *_):
            assert name == "elfihip_kept_rows"
            ep._obj.value = self.epoch
    ctx, other = FakeCtx(), FakeCtx()
    a = np.zeros((5, 2))
    ctx.epoch = 7
    assert _lib.remember_kept(a, ctx) is a
    assert _lib.kept_epoch_of(a, ctx) == 7 and _lib.kept_epoch_of(a, other) is None
    assert _lib.kept_epoch_of(a.copy(), ctx) is None
    v = a.reshape(-1)
    assert _lib.alias_kept(v, a) is v and _lib.kept_epoch_of(v, ctx) == 7
    b = np.ones(3)
    assert _lib.alias_kept(b.reshape(-1), b) is not None and _lib.kept_epoch_of(b, ctx) is None      # nothing to alias
    # the hand-over is exact: what names a device copy is read-only, a copy is an ordinary array that takes the upload
    # path, and an array somebody made writeable again no longer names the device copy
    with pytest.raises(ValueError):
        a[0, 0] = 1.0
    with pytest.raises(ValueError):
        v[3] = 1.0
    a2 = a.copy()
    a2[0, 0] = 1.0
    assert a2.flags.writeable and _lib.kept_epoch_of(a2, ctx) is None and b.flags.writeable
    w = np.zeros(4)
    _lib.remember_kept(w, ctx)
    w.flags.writeable = True
    w[np.array([2])] = np.inf              # the scattered edit the round-5 fingerprint could miss
    assert _lib.kept_epoch_of(w, ctx) is None and id(w) not in _lib._KEPT
    prev = _lib.set_device_handover(False)
    try:
        free = _lib.remember_kept(np.zeros(3), ctx)
        assert free.flags.writeable and _lib.kept_epoch_of(free, ctx) is None
    finally:
        _lib.set_device_handover(prev)
    ident = id(a)
    del a, v
    gc.collect()
    assert ident not in _lib._KEPT or _lib._KEPT[ident][0]() is None
    for i in range(100):                        # the registry stays small
        _lib.remember_kept(np.zeros(1), ctx)
    assert len(_lib._KEPT) <= 33
    r = np.zeros((4, 6))
    ctx.epoch = 9
    assert _lib.remember_rows(r, ctx) is r and _lib.rows_epoch_of(r, ctx) == 9 and _lib.rows_epoch_of(r, other) is None
    with pytest.raises(ValueError):
        r *= 2.0
    r.flags.writeable = True
    assert _lib.rows_epoch_of(r, ctx) is None
    p = _lib.pinned.array((10, 4))
    assert p.shape == (10, 4) and p.dtype == np.float64 and p.flags['WRITEABLE']
    p[:] = 3.0
    assert float(p.sum()) == 120.0


def test_prior_and_proposal_argument_checks_need_no_gpu():
    """elfi_amd.priors / GMDistribution.rvs validate their arguments before any device call (shape errors are ValueErrors,
    as scipy's are), and the densities of the prior objects are the reference distributions' own."""
    import scipy.stats as ss
    from elfi_amd import priors
    from elfi_amd.gmix import GMDistribution
    with pytest.raises(ValueError):
        priors.prior_draw(priors.UNIFORM, (1.0,), None, (4,), seed=1)          # loc AND scale
    with pytest.raises(ValueError):
        priors.prior_draw(priors.MA2_T1, (1.0, 2.0), None, (4,), seed=1)       # b only
    assert priors._shape(None) == () and priors._shape(5) == (5,) and priors._shape((2, 3)) == (2, 3)
    x = np.linspace(-1, 6, 9)
    assert np.array_equal(priors.uniform.pdf(x, 1.0, 4.0), ss.uniform.pdf(x, 1.0, 4.0))
    assert np.array_equal(priors.uniform.logpdf(x, 1.0, 4.0), ss.uniform.logpdf(x, 1.0, 4.0))
    assert np.array_equal(priors.uniform.cdf(x, 1.0, 4.0), ss.uniform.cdf(x, 1.0, 4.0))
    # array parameters are refused (no silent host path)
    with pytest.raises(ValueError):
        priors.uniform.rvs(np.zeros(3), np.ones(3), size=3, random_state=np.random.RandomState(0))
    with pytest.raises(ValueError):
        GMDistribution.rvs(np.zeros((4, 3)), np.eye(2), None, size=5)          # covariance of the wrong dimension
    with pytest.raises(NotImplementedError):
        GMDistribution.rvs(np.zeros((4, 65)), 1.0, None, size=5)


def test_model_prior_replay_plan_is_the_public_call_bit_for_bit():
    """lcb_acquisition.prior_rvs: ModelPrior.rvs (elfi/model/extensions.py:156-174) replayed from a plan made once --
    same draws and same generator state as the public call, call after call, for dependent priors (MA2), many
    independent ones and a one-dimensional prior; objects that are not a ModelPrior take the public call."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
    import ref_shim
    if not ref_shim.available():
        pytest.skip('reference ELFI not available')
    ref_shim.install()
    import scipy.stats as ss
    import elfi
    from elfi.examples import ma2
    from elfi.model.extensions import ModelPrior
    from elfi_amd import lcb_acquisition as L

    def same_stream(mp, n, reps=5):
        a, b = np.random.RandomState(7), np.random.RandomState(7)
        for _ in range(reps):
            x, y = mp.rvs(n, random_state=a), L.prior_rvs(mp, n, b)
            assert np.shape(x) == np.shape(y) and np.array_equal(x, y)
            sa, sb = a.get_state(), b.get_state()
            assert sa[0] == sb[0] and np.array_equal(sa[1], sb[1]) and sa[2:] == sb[2:]
        from elfi_amd import elfi_plans
        return elfi_plans._PLANS[mp][('rvs', n)]

    mp = ModelPrior(ma2.get_model(seed_obs=1))
    from elfi_amd.elfi_plans import NetPlan
    assert isinstance(same_stream(mp, 10), NetPlan) and isinstance(same_stream(mp, 3), NetPlan)
    m2 = elfi.new_model()
    for i in range(6):
        elfi.Prior(ss.uniform, -2, 4, model=m2, name='p%d' % i)
    assert isinstance(same_stream(ModelPrior(m2), 64), NetPlan)
    m3 = elfi.new_model()
    elfi.Prior('norm', 1, 2, model=m3, name='a')
    mp3 = ModelPrior(m3)
    same_stream(mp3, 5)
    assert L.prior_rvs(mp3, 5, np.random.RandomState(1)).shape == (5,)
    # the start points of a search come through it, clipped to the bounds as the reference clips them
    pts = L.draw_start_points([(-2, 2), (-1, 1)], 10, mp, np.random.RandomState(3))
    ref = np.clip(mp.rvs(10, random_state=np.random.RandomState(3)), [-2, -1], [2, 1])
    assert np.array_equal(pts, ref)

    class Other:
        def rvs(self, n, random_state=None):
            return np.zeros((n, 2))
    assert L.prior_rvs(Other(), 4, np.random.RandomState(0)).shape == (4, 2)


def test_batched_prior_gradient_is_the_references_row_by_row():
    """posterior.prior_logpdf_and_gradient: ModelPrior.logpdf / gradient_logpdf (extensions.py:176-240, numgrad
    utils.py:275-314) for all rows from ONE pass over their stencils -- equal to the reference's row loop value for
    value, inside the support, outside it, on its edge; other prior objects keep their own methods."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
    import ref_shim
    if not ref_shim.available():
        pytest.skip('reference ELFI not available')
    ref_shim.install()
    import elfi
    from elfi.examples import ma2
    from elfi.model.extensions import ModelPrior
    from elfi_amd import posterior as P
    rs = np.random.RandomState(0)
    mp = ModelPrior(ma2.get_model(seed_obs=1))
    x = np.vstack([mp.rvs(9, random_state=rs), [[3.0, 0.0], [0.0, 2.0], [1.99999, 0.9], [-2.0, 0.0]]])
    lp, g = P.prior_logpdf_and_gradient(mp, x)
    assert np.array_equal(lp, mp.logpdf(x)) and np.array_equal(g, mp.gradient_logpdf(x))
    m2 = elfi.new_model()
    elfi.Prior('norm', 1, 2, model=m2, name='a')
    elfi.Prior('gamma', 2.0, model=m2, name='b')
    elfi.Prior('beta', 2.0, 3.0, model=m2, name='c')
    mp2 = ModelPrior(m2)
    x2 = np.column_stack([rs.randn(11) + 1, np.abs(rs.randn(11)) - 0.2, rs.uniform(-0.1, 1.1, 11)])
    lp2, g2 = P.prior_logpdf_and_gradient(mp2, x2)
    assert np.array_equal(lp2, mp2.logpdf(x2)) and np.array_equal(g2, mp2.gradient_logpdf(x2))
    assert np.any(g2 != 0.0) and np.any(np.isneginf(lp2))
    # the pass itself comes from a plan after the first (verified) use, for the density as for its logarithm
    from elfi_amd import elfi_plans as E
    for log in (True, False):
        public = mp2.logpdf if log else mp2.pdf
        for _ in range(3):
            assert np.array_equal(E.prior_logpdf(mp2, x2, log), public(x2), equal_nan=True)
        assert isinstance(E._PLANS[mp2][('logpdf' if log else 'pdf', len(x2))], E.NetPlan)
    assert np.array_equal(E.prior_logpdf(mp2, x2[:4]), mp2.logpdf(x2[:4]))       # another batch size: another plan
    assert np.array_equal(E.prior_logpdf(mp2, x2[0]), mp2.logpdf(x2[0]))         # not (n, dim): the public call
    box = P._UniformBoxPrior([(-1, 1), (0, 2)])
    xb = rs.uniform(-2, 3, (7, 2))
    lb, gb = P.prior_logpdf_and_gradient(box, xb)
    assert np.array_equal(lb, box.logpdf(xb)) and np.array_equal(gb, box.gradient_logpdf(xb))
