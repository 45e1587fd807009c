"""GPU: prior draws on the device (elfi_amd/priors.py, csrc/gauss.hip prior_draw_kernel).

The uniforms are the library's own (Philox4x32-10), so the parity statement is about the TRANSFORM: on the same uniforms
the device results equal the reference's NumPy expressions bit for bit (elfi/examples/ma2.py:102-118,147-166;
ss.uniform.rvs = U * scale + loc), the uniforms are the 53-bit lattice in [0, 1) made of the generator's words
(tests/philox_ref.py), and the draws are distributed as the reference's priors.
"""
import numpy as np
import pytest
import scipy.stats as ss

pytestmark = pytest.mark.gpu


def _uniforms(n, seed, stream=0):
    from elfi_amd import priors
    return priors.prior_draw(priors.UNIFORM, (0.0, 1.0), None, (n,), seed=seed, stream=stream)


@pytest.mark.parametrize('n', [1, 2, 7, 1000, 100001])
def test_uniforms_are_the_generators_words(hip_ctx, n):
    import philox_ref as P
    u = _uniforms(n, seed=12345, stream=3)
    assert u.shape == (n,) and np.all(u >= 0.0) and np.all(u < 1.0)
    for e in sorted({0, min(1, n - 1), n // 2, n - 1}):
        r = P.philox4x32_10(((e // 2) & 0xFFFFFFFF, (e // 2) >> 32, 3, 0), (12345, 0))
        hi, lo = (r[0], r[1]) if e % 2 == 0 else (r[2], r[3])
        assert u[e] == ((hi << 21) | (lo >> 11)) * 2.0 ** -53
    assert np.array_equal(u, _uniforms(n, seed=12345, stream=3))
    if n > 1:
        assert not np.array_equal(u, _uniforms(n, seed=12346, stream=3))


def test_transforms_are_numpys_bit_for_bit(hip_ctx):
    from elfi_amd import priors
    n = 200003
    u = _uniforms(n, seed=7)
    # ss.uniform.rvs(loc, scale): U * scale + loc
    got = priors.prior_draw(priors.UNIFORM, (-1.25, 3.5), None, (n,), seed=7)
    assert np.array_equal(got, u * 3.5 + -1.25)
    # CustomPrior1.rvs(b)
    b = 2.0
    t1 = priors.prior_draw(priors.MA2_T1, (b,), None, (n,), seed=7)
    assert np.array_equal(t1, np.where(u < 0.5, np.sqrt(2. * u) * b - b, -np.sqrt(2. * (1. - u)) * b + b))
    # CustomPrior2.rvs(t1, a)
    a = 1.0
    t2 = priors.prior_draw(priors.MA2_T2, (a,), t1, (n,), seed=7)
    locs = np.maximum(-a - t1, -a + t1)
    assert np.array_equal(t2, u * (a - locs) + locs)
    assert np.all(np.abs(t1) <= b) and np.all(t2 <= a) and np.all(t2 >= locs)


def test_distribution_objects_behave_like_the_references(hip_ctx):
    """rvs shapes / seeding from the node's RandomState, densities delegated, distributions by Kolmogorov-Smirnov."""
    import ref_shim
    if not ref_shim.available():
        pytest.skip('reference ELFI not available')
    ref_shim.install()
    from elfi.examples import ma2
    from elfi_amd import priors
    rs = np.random.RandomState(3)
    t1 = priors.MA2Prior1.rvs(2, size=(50000,), random_state=rs)
    t2 = priors.MA2Prior2.rvs(t1, 1, size=(50000,), random_state=rs)
    assert t1.shape == t2.shape == (50000,)
    # same RandomState seed -> same draws; the state advances by one randint per call
    rs2 = np.random.RandomState(3)
    assert np.array_equal(priors.MA2Prior1.rvs(2, size=(50000,), random_state=rs2), t1)
    assert rs2.randint(10 ** 9) != np.random.RandomState(3).randint(10 ** 9)
    # the triangle of Marin et al.: t1 in [-2, 2], t1 + t2 > -1, t1 - t2 < 1, |t2| < 1
    assert np.all(t1 + t2 >= -1) and np.all(t1 - t2 <= 1) and np.all(np.abs(t2) <= 1)
    r1 = ma2.CustomPrior1.rvs(2, size=(50000,), random_state=np.random.RandomState(5))
    assert ss.ks_2samp(t1, r1).pvalue > 1e-3
    r2 = ma2.CustomPrior2.rvs(r1, 1, size=(50000,), random_state=np.random.RandomState(6))
    assert ss.ks_2samp(t2, r2).pvalue > 1e-3
    x = np.linspace(-2.5, 2.5, 11)
    assert np.array_equal(priors.MA2Prior1.pdf(x, 2), ma2.CustomPrior1.pdf(x, 2))
    assert np.array_equal(priors.MA2Prior2.pdf(x, 0.3, 1), ma2.CustomPrior2.pdf(x, 0.3, 1))
    u = priors.uniform.rvs(1.0, 4.0, size=(3, 5), random_state=np.random.RandomState(1))
    assert u.shape == (3, 5) and np.all(u >= 1.0) and np.all(u < 5.0)
    assert priors.uniform.pdf(2.0, 1.0, 4.0) == 0.25
    assert np.ndim(priors.uniform.rvs(0, 1, size=None, random_state=np.random.RandomState(1))) == 0


def test_ma2_model_with_device_priors_through_the_reference_loop(hip_ctx):
    import ref_shim
    if not ref_shim.available():
        pytest.skip('reference ELFI not available')
    ref_shim.install()
    import elfi
    import elfi_amd
    m = elfi_amd.fused_models.ma2_model(n_obs=100, seed_obs=4)
    res = elfi_amd.HipRejection(m['d'], batch_size=20000, seed=2).sample(200, n_sim=100000)
    t1, t2 = res.samples['t1'], res.samples['t2']
    assert t1.shape == (200,) and np.all(np.isfinite(res.discrepancies))
    # the posterior concentrates around the generating parameters (0.6, 0.2)
    assert abs(np.mean(t1) - 0.6) < 0.25 and abs(np.mean(t2) - 0.2) < 0.25
    # the reference's own sampler accepts the same model
    res2 = elfi.Rejection(m['d'], batch_size=20000, seed=2).sample(200, n_sim=100000)
    assert np.array_equal(res2.samples['t1'], t1) and np.array_equal(res2.discrepancies, res.discrepancies)
