"""MaxVar / RandMaxVar on the device GP against the reference's own classes (tests/golden/maxvar.npz)."""
import os

import numpy as np
import pytest

import posterior_oracle as PO
from conftest import GOLDEN
from test_maxvar import check_against_fixture, check_expintvar, check_gradient_numerically

pytestmark = pytest.mark.gpu


def _device_model(g):
    import elfi_amd
    d = g['X'].shape[1]
    names = ['p%d' % i for i in range(d)]
    m = elfi_amd.HipGPRegression(names, bounds=dict(zip(names, [tuple(b) for b in g['bounds']])))
    m.update(g['X'], g['y'])
    m._hyper = dict(zip(('var', 'ls', 'bias', 'noise'), (float(v) for v in g['hyper'])))
    m._refit()
    return m


def test_maxvar_on_the_device_gp_equals_the_reference(hip_ctx):
    g, model, prior = check_against_fixture(_device_model, 1e-7, 1e-4)
    assert model._handle is not None, 'the device GP must be the one that ran'


def test_randmaxvar_on_the_device_gp(hip_ctx):
    import elfi_amd
    g = np.load(os.path.join(GOLDEN, 'maxvar.npz'))
    model = _device_model(g)
    prior = PO.BoxPrior(model.bounds)
    lo, hi = np.array(model.bounds).T
    for sampler in ('nuts', 'metropolis'):
        r1 = elfi_amd.HipRandMaxVar(model, prior, quantile_eps=0.05, sampler=sampler, n_samples=40, seed=9)
        a = r1.acquire(1)
        # same random stream and algorithm; device and host evaluations differ in the last bits, which a chain
        # amplifies: the short Metropolis chain stays on the reference's path, NUTS is compared as a sample
        assert a.shape == (1, 2) and np.all(a >= lo) and np.all(a <= hi)
        if sampler == 'metropolis':
            np.testing.assert_allclose(a, g['rand_metropolis_1'], rtol=0, atol=1e-5)
        r3 = elfi_amd.HipRandMaxVar(model, prior, quantile_eps=0.05, sampler=sampler, n_samples=40, seed=9)
        b = r3.acquire(3)
        assert b.shape == (3, 2) and np.all(b >= lo) and np.all(b <= hi)
        again = elfi_amd.HipRandMaxVar(model, prior, quantile_eps=0.05, sampler=sampler, n_samples=40, seed=9).acquire(3)
        assert np.array_equal(b, again)


def test_expintvar_on_the_device_gp(hip_ctx):
    g, ev, ei, thi = check_expintvar(_device_model, 1e-7, 1e-4)
    scale = np.max(np.abs(g['eiv_imp_loss']))
    assert thi.shape == (1, 2) and np.all(np.abs(thi) <= 2)
    assert ei.points_int.shape == g['eiv_imp_points'].shape
    # posterior covariance against the point set: the device answer vs the NumPy formula of the oracle model
    from test_maxvar import OracleModel
    om = OracleModel(g)
    om.set_integration_points(ev.points_int)
    model = ev.model
    model.set_integration_points(ev.points_int)
    cov, var = model.cross_cov(g['xs'])
    cov_ref, var_ref = om.cross_cov(g['xs'])
    np.testing.assert_allclose(cov, cov_ref, rtol=0, atol=1e-10 * max(1.0, np.max(np.abs(cov_ref))))
    np.testing.assert_allclose(var, var_ref, rtol=1e-8, atol=1e-12)
    # the point set belongs to one factorisation: new evidence invalidates it loudly
    model.update(np.array([[0.1, 0.2]]), np.array([[0.5]]))
    with pytest.raises(RuntimeError):
        model.cross_cov(g['xs'])


def test_maxvar_gradient_is_the_derivative_of_the_value_on_the_device(hip_ctx):
    g = np.load(os.path.join(GOLDEN, 'maxvar.npz'))
    check_gradient_numerically(_device_model(g))
