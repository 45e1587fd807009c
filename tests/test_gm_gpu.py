"""GPU: Gaussian-mixture proposal density vs the reference's GMDistribution (golden fixture from the
real class) and the oracle.  Tolerance 1e-12 relative: same factorisation of the covariance and same
accumulation order as the reference; the device exp() differs from libm's in the last bits."""
import os

import numpy as np
import pytest

import gm_oracle as GM
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def test_against_the_reference_class(hip_ctx):
    import elfi_amd
    g = np.load(os.path.join(GOLDEN, 'gm_pdf.npz'))
    for k in g['cases']:
        x, means, w = g['x_%d' % k], g['means_%d' % k], g['w_%d' % k]
        cov = g['cov_%d' % k]
        cov = float(cov) if cov.ndim == 0 else cov
        got = elfi_amd.GMDistribution.pdf(x, means, cov=cov, weights=w)
        np.testing.assert_allclose(got, g['pdf_%d' % k], rtol=1e-12, atol=1e-300)
        np.testing.assert_allclose(elfi_amd.GMDistribution.logpdf(x, means, cov=cov, weights=w), g['logpdf_%d' % k],
                                   rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(elfi_amd.GMDistribution.pdf(x, means, cov=cov), g['pdf_now_%d' % k], rtol=1e-12,
                                   atol=1e-300)
    single = elfi_amd.GMDistribution.pdf(g['x_1'][3], g['means_1'], cov=g['cov_1'], weights=g['w_1'])
    assert np.ndim(single) == 0 and single == pytest.approx(float(g['pdf_single']), rel=1e-12)


@pytest.mark.parametrize('M,N,d', [(1, 1, 1), (1000, 1000, 2), (5000, 257, 4), (300, 2000, 8), (77, 50, 16)])
def test_vs_oracle(hip_ctx, M, N, d):
    import elfi_amd
    rs = np.random.RandomState(M + N + d)
    means = rs.randn(N, d)
    x = rs.randn(M, d) * 1.5
    A = rs.randn(d, d)
    cov = (A @ A.T + d * np.eye(d)) / (3.0 * d)
    w = rs.uniform(0.5, 1.5, N)
    got = elfi_amd.GMDistribution.pdf(x, means, cov=cov, weights=w)
    np.testing.assert_allclose(got, GM.pdf(x, means, cov=cov, weights=w), rtol=1e-12, atol=1e-300)
    got = elfi_amd.GMDistribution.pdf(x, means, cov=0.37)              # scalar covariance, unit weights
    np.testing.assert_allclose(got, GM.pdf(x, means, cov=0.37), rtol=1e-12, atol=1e-300)


def test_errors(hip_ctx):
    import elfi_amd
    with pytest.raises(ValueError):
        elfi_amd.GMDistribution.pdf(np.zeros((3, 2)), np.zeros((4, 3)))
    with pytest.raises(np.linalg.LinAlgError):
        elfi_amd.GMDistribution.pdf(np.zeros((3, 2)), np.zeros((4, 2)), cov=np.ones((2, 2)))
    with pytest.raises(NotImplementedError):
        elfi_amd.GMDistribution.pdf(np.zeros((3, 65)), np.zeros((4, 65)))


@pytest.mark.parametrize('d', [17, 20, 32, 33, 64])
def test_many_parameters(hip_ctx, d):
    """16 < d <= 64: both point sets are transformed by the covariance factor once, a pair then costs d subtractions."""
    import elfi_amd
    import gm_oracle as GM
    rs = np.random.RandomState(d)
    M, N = 777, 300
    x, means = rs.randn(M, d), rs.randn(N, d) * 0.8
    A = rs.randn(d, d)
    cov = A @ A.T / d + 0.5 * np.eye(d)
    w = rs.uniform(0.5, 1.5, N)
    ref = GM.pdf(x, means, cov=cov, weights=w)
    np.testing.assert_allclose(elfi_amd.GMDistribution.pdf(x, means, cov=cov, weights=w), ref, rtol=1e-11, atol=1e-300)
    np.testing.assert_allclose(elfi_amd.GMDistribution.logpdf(x, means, cov=cov, weights=w), np.log(ref), rtol=1e-11, atol=1e-11)


def test_rvs_on_the_device_has_the_mixtures_moments(hip_ctx):
    """GMDistribution.rvs (elfi/methods/utils.py:199-262) on the device: shapes as the reference's, components by their
    weights, perturbations with the shared covariance, the validity loop against the prior; seeded by the RandomState."""
    from elfi_amd import GMDistribution
    rs = np.random.RandomState(0)
    N, d, n = 7, 3, 400000
    means = rs.uniform(-5, 5, (N, d))
    w = rs.uniform(0.1, 1.0, N)
    w[2] = 0.0                                     # a component without mass is never drawn
    B = rs.randn(d, d)
    cov = B @ B.T / d + 0.1 * np.eye(d)
    x = GMDistribution.rvs(means, cov, w, size=n, random_state=np.random.RandomState(5))
    assert x.shape == (n, d)
    assert np.array_equal(x, GMDistribution.rvs(means, cov, w, size=n, random_state=np.random.RandomState(5)))
    assert not np.array_equal(x, GMDistribution.rvs(means, cov, w, size=n, random_state=np.random.RandomState(6)))
    wn = w / w.sum()
    m_true = wn @ means
    c_true = cov + (means - m_true).T @ np.diag(wn) @ (means - m_true)
    assert np.max(np.abs(x.mean(axis=0) - m_true)) < 5 * np.sqrt(np.max(np.diag(c_true)) / n) + 1e-12
    assert np.max(np.abs(np.cov(x.T) - c_true)) < 0.02 * np.max(np.abs(c_true))
    # well separated components: the shares of the draws are the weights, a component without mass is never drawn
    far = 100.0 * np.arange(N)[:, None] * np.ones((1, d))
    xf = GMDistribution.rvs(far, 0.01 * np.eye(d), w, size=n, random_state=np.random.RandomState(7))
    comp = np.rint(xf[:, 0] / 100.0).astype(int)
    share = np.bincount(comp, minlength=N) / n
    assert share[2] == 0.0
    assert np.max(np.abs(share - wn)) < 5 * np.sqrt(0.25 / n)
    # 1-d means, scalar covariance, size None / int
    m1 = np.array([-2.0, 0.5, 3.0])
    y = GMDistribution.rvs(m1, 0.25, None, size=200000, random_state=np.random.RandomState(1))
    assert y.shape == (200000,)
    assert abs(y.mean() - m1.mean()) < 0.02 and abs(y.var() - (0.25 + m1.var())) < 0.05
    one = GMDistribution.rvs(m1, 0.25, None, size=None, random_state=np.random.RandomState(1))
    assert np.ndim(one) == 0
    # the validity loop: only points the prior accepts come back, still `size` of them
    logp = lambda v: np.where(v > 0.0, 0.0, -np.inf)     # noqa: E731
    z = GMDistribution.rvs(m1, 0.25, None, size=50000, prior_logpdf=logp, random_state=np.random.RandomState(2))
    assert z.shape == (50000,) and np.all(z > 0.0)
    with pytest.raises(ValueError):
        GMDistribution.rvs(means, np.eye(2), w, size=3)
