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
