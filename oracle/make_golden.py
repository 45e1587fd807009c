"""TEST INFRASTRUCTURE ONLY -- regenerates tests/golden/*.npz by running REAL reference ELFI.

Run in the build container (needs /root/reference; the GPU box never runs this):

    python oracle/make_golden.py            # all fixtures
    python oracle/make_golden.py ma2 adaptive metrics

Fixtures (inputs + the reference's outputs on them):

  ma2_tutorial.npz   docs/usage/tutorial.rst:28-29,95-100,290-323,360,386,396 -- the
                     documented run Rejection(d, batch_size=10000, seed=20170530)
                     .sample(1000, quantile=0.01) -> threshold 0.116859716394976.
                     Holds every batch's summaries S1, S2 (inputs of the Distance
                     operation), the distances the reference computed, the observed
                     summaries, and the final threshold / sample means.
  adaptive_ex1.npz   docs/usage/adaptive_distance.rst:43-214 (weights [0.06940134, 0.0097677])
  adaptive_ex2.npz   docs/usage/adaptive_distance.rst:231-378 (seven weight vectors)
                     -- full call trace of the AdaptiveDistance node (add_data /
                     update_distance / nested_distance) recorded from the reference run.
  metrics.npz        cdist through elfi.Distance's own partial(...) for every metric /
                     keyword form the node accepts, on seeded random summaries.

Outputs larger than a few hundred rows are stored as a leading slice plus a SHA-256 of
the full float64 byte string (the GPU kernels are bit-exact on these metrics, so the
parity tests compare digests as well as the slices).
"""
import hashlib
import os
import sys

import numpy as np
import scipy.stats as ss

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float64).tobytes()).hexdigest()


class Tap:
    """Record every call of a node operation (args, kwargs, result)."""

    def __init__(self, op, log):
        self.op, self.log = op, log

    def __call__(self, *a, **k):
        out = self.op(*a, **k)
        self.log.append((a, k, out))
        return out


def make_ma2(elfi):
    from elfi.examples.ma2 import MA2, autocov
    from elfi.examples.ma2 import CustomPrior1 as CustomPrior_t1, CustomPrior2 as CustomPrior_t2
    seed = 20170530
    np.random.seed(seed)
    y_obs = MA2(0.6, 0.2)
    m = elfi.new_model()
    t1 = elfi.Prior(CustomPrior_t1, 2, model=m, name='t1')
    t2 = elfi.Prior(CustomPrior_t2, t1, 1, name='t2')
    Y = elfi.Simulator(MA2, t1, t2, observed=y_obs, name='MA2')
    S1 = elfi.Summary(autocov, Y, name='S1')
    S2 = elfi.Summary(autocov, Y, 2, name='S2')
    d = elfi.Distance('euclidean', S1, S2, name='d')
    log = []
    d.state['attr_dict']['_operation'] = Tap(d.state['attr_dict']['_operation'], log)
    rej = elfi.Rejection(d, batch_size=10000, seed=seed, output_names=['S1', 'S2'])
    res = rej.sample(1000, quantile=0.01)
    assert repr(float(res.threshold)) == '0.116859716394976', repr(res.threshold)
    # first logged call is on the observed data?  no: observed summaries are computed by the
    # compiler and handed in through `observed=`; every logged call is one batch.
    S1b = np.stack([c[0][0] for c in log])
    S2b = np.stack([c[0][1] for c in log])
    dist = np.stack([c[2] for c in log])
    obs = log[0][1]['observed']
    np.savez_compressed(
        os.path.join(GOLDEN, 'ma2_tutorial.npz'),
        S1=S1b, S2=S2b, d0=dist[0], d_sha=np.array(sha(dist)), observed=np.concatenate([np.atleast_2d(o) for o in obs], axis=1),
        y_obs=y_obs, threshold=np.float64(res.threshold),
        mean_t1=np.float64(res.sample_means['t1']), mean_t2=np.float64(res.sample_means['t2']),
        n_sim=np.int64(res.n_sim))
    print('ma2_tutorial: batches', S1b.shape, 'threshold', repr(float(res.threshold)))


def make_gauss(elfi):
    """The Gaussian example (elfi/examples/gauss.py) through the reference's own functions: simulator output for known
    parameters and a known RandomState, both summaries, and the Distance node's operation on them."""
    from functools import partial
    import scipy.spatial.distance
    from elfi.examples import gauss as G
    from elfi.model.utils import distance_as_discrepancy
    out = {}
    for tag, B, n_obs, seed in (('a', 300, 50, 11), ('b', 257, 7, 12), ('c', 48, 200, 13), ('d', 33, 129, 14)):
        rs = np.random.RandomState(seed)
        mu = rs.uniform(-1, 9, B)
        sigma = rs.uniform(0.01, 5.0, B)
        draw_seed = 1000 + seed
        y = G.gauss(mu, sigma, n_obs=n_obs, batch_size=B, random_state=np.random.RandomState(draw_seed))
        y_obs = G.gauss(4, 0.4, n_obs=n_obs, random_state=np.random.RandomState(seed + 50))
        sm, sv = G.ss_mean(y), G.ss_var(y)
        om, ov = G.ss_mean(y_obs), G.ss_var(y_obs)
        d = distance_as_discrepancy(partial(scipy.spatial.distance.cdist, metric='euclidean'), sm, sv, observed=(om, ov))
        out.update({'mu_' + tag: mu, 'sigma_' + tag: sigma, 'y_' + tag: y, 'ss_mean_' + tag: sm, 'ss_var_' + tag: sv,
                    'observed_' + tag: np.array([om[0], ov[0]]), 'd_' + tag: d,
                    'draw_seed_' + tag: np.int64(draw_seed), 'n_obs_' + tag: np.int64(n_obs)})
    out['cases'] = np.array(['a', 'b', 'c', 'd'])
    # the whole example model once: what the nodes hand to one another in a batch
    m = G.get_model(n_obs=50, seed_obs=3)
    batch = m.generate(200, outputs=['mu', 'sigma', 'gauss', 'ss_mean', 'ss_var', 'd'], seed=5)
    for k, v in batch.items():
        out['model_' + k] = v
    out['model_observed'] = np.array([G.ss_mean(m.observed['gauss'])[0], G.ss_var(m.observed['gauss'])[0]])
    np.savez_compressed(os.path.join(GOLDEN, 'gauss_example.npz'), **out)
    print('gauss_example:', [out['y_' + t].shape for t in out['cases']], 'model batch', batch['gauss'].shape)


def _trace_adaptive(elfi, simulator, prior, observed, batch_size, seed, calls, tag):
    """Run AdaptiveDistanceSMC on the real reference with a recording AdaptiveDistance."""
    events = []

    class RecAdaptiveDistance(elfi.AdaptiveDistance):
        def add_data(self, *data):
            events.append(('add_data', np.column_stack(data).copy()))
            return super().add_data(*data)

        def update_distance(self):
            super().update_distance()
            events.append(('update_distance', self.state['w'][-1].copy()))

        def nested_distance(self, u, v):
            out = super().nested_distance(u, v)
            events.append(('nested_distance', u.copy(), v.copy(), out.copy()))
            return out

    m = elfi.new_model()
    theta = elfi.Prior(*prior, model=m, name='theta')
    sim = elfi.Simulator(simulator, theta, observed=observed, name='sim')
    d = RecAdaptiveDistance(sim, name='d')
    ada = elfi.AdaptiveDistanceSMC(d, batch_size=batch_size, seed=seed)
    results = [ada.sample(*c[0], **c[1]) for c in calls]
    out = {}
    kinds = []
    for i, ev in enumerate(events):
        kinds.append(ev[0])
        if ev[0] == 'add_data':
            out['e%d_data' % i] = ev[1]
        elif ev[0] == 'update_distance':
            out['e%d_w' % i] = ev[1]
        else:
            out['e%d_u' % i] = ev[1]
            out['e%d_v' % i] = ev[2]
            out['e%d_head' % i] = ev[3][:64]
            out['e%d_sha' % i] = np.array(sha(ev[3]))
            out['e%d_shape' % i] = np.array(ev[3].shape)
    out['kinds'] = np.array(kinds)
    last = results[-1]
    out['final_w'] = np.array(last.adaptive_distance_w)
    out['threshold'] = np.float64(last.threshold)
    out['n_sim'] = np.int64(last.n_sim)
    out['mean_theta'] = np.float64(last.sample_means['theta'])
    np.savez_compressed(os.path.join(GOLDEN, tag + '.npz'), **out)
    print(tag, 'events', len(events), 'w', out['final_w'].round(8).tolist(), 'thr', float(last.threshold))
    return last


def make_adaptive(elfi):
    def simulator1(mu, batch_size=1, random_state=None):
        mu = np.asarray(mu).reshape((-1, 1))
        o1 = ss.norm.rvs(loc=mu, scale=1, random_state=random_state).reshape((-1, 1))
        o2 = ss.norm.rvs(loc=mu, scale=100, random_state=random_state).reshape((-1, 1))
        return np.hstack((o1, o2))

    r1 = _trace_adaptive(elfi, simulator1, (ss.uniform, 0, 50), np.array([20, 20])[None, :],
                         10000, 123, [((100, 1), dict(quantile=0.01))], 'adaptive_ex1')
    assert np.allclose(r1.adaptive_distance_w[0], [0.06940134, 0.0097677], rtol=0, atol=5e-9)

    def simulator2(mu, batch_size=1, random_state=None):
        mu = np.asarray(mu).reshape((-1, 1))
        o1 = ss.norm.rvs(loc=mu, scale=0.1, random_state=random_state).reshape((-1, 1))
        o2 = ss.norm.rvs(loc=1, scale=1, size=batch_size, random_state=random_state).reshape((-1, 1))
        return np.hstack((o1, o2))

    r2 = _trace_adaptive(elfi, simulator2, (ss.norm, 0, 100), np.array([0, 0])[None, :],
                         2000, 123, [((1000, 5), {}), ((1000, 2), {})], 'adaptive_ex2')
    doc = np.array([[0.01023228, 1.00584519], [0.00921258, 0.99287166], [0.01201937, 0.99365522],
                    [0.02217631, 0.98925365], [0.04355987, 1.00076738], [0.07863284, 0.9971017],
                    [0.13892778, 1.00929049]])
    assert np.allclose(np.array(r2.adaptive_distance_w), doc, rtol=0, atol=5e-9)


def make_metrics(elfi):
    """Every metric / keyword form through the partial() elfi.Distance itself builds."""
    rs = np.random.RandomState(20240917)
    out = {}
    cases = []
    for m in (1, 2, 3, 32, 33, 64, 100):
        n = 389 if m < 64 else 131
        X = rs.randn(n, m) * rs.uniform(0.1, 10, m)
        y = rs.randn(1, m)
        w = rs.uniform(0.5, 2.0, m)
        w0 = w.copy()
        w0[::3] = 0.0
        A = rs.randn(m, m)
        VI = A @ A.T + m * np.eye(m)
        out['X_%d' % m], out['y_%d' % m], out['w_%d' % m] = X, y, w
        out['w0_%d' % m], out['VI_%d' % m] = w0, VI
        forms = [('euclidean', {}), ('euclidean', dict(w=w)), ('sqeuclidean', {}),
                 ('sqeuclidean', dict(w=w)), ('cityblock', {}), ('cityblock', dict(w=w)),
                 ('chebyshev', {}), ('chebyshev', dict(w=w0)), ('minkowski', dict(p=1)),
                 ('minkowski', dict(p=2)), ('minkowski', dict(p=3)), ('minkowski', dict(p=2.5, w=w)),
                 ('minkowski', dict(p=np.inf)), ('seuclidean', dict(V=w)),
                 ('mahalanobis', dict(VI=VI))]
        for k, (metric, kw) in enumerate(forms):
            mdl = elfi.new_model()
            c0 = elfi.Constant(0, model=mdl, name='c0')
            S = elfi.Summary(lambda x: x, c0, observed=y, model=mdl, name='S')
            node = elfi.Distance(metric, S, model=mdl, name='d', **dict(kw))
            op = node.state['attr_dict']['_operation']  # partial(distance_as_discrepancy, partial(cdist, ...))
            res = op(X, observed=(y,))
            key = 'd_%d_%d' % (m, k)
            out[key] = res
            cases.append('%d|%d|%s|%s' % (m, k, metric, ','.join(
                '%s=%s' % (a, ('w0' if (a == 'w' and metric == 'chebyshev') else a)
                           if a in ('w', 'V', 'VI') else repr(float(b))) for a, b in kw.items())))
    out['cases'] = np.array(cases)
    np.savez_compressed(os.path.join(GOLDEN, 'metrics.npz'), **out)
    print('metrics:', len(cases), 'cases')


def make_gm(elfi):
    """GMDistribution.pdf / logpdf of the real reference (elfi/methods/utils.py:139-198)."""
    from elfi.methods.utils import GMDistribution
    rs = np.random.RandomState(777)
    out, cases = {}, []
    for k, (M, N, d) in enumerate([(50, 7, 1), (200, 40, 2), (300, 120, 3), (120, 300, 5), (64, 33, 10)]):
        means = rs.randn(N, d) * 2.0
        x = np.vstack([means[rs.randint(0, N, M // 2)] + 0.3 * rs.randn(M // 2, d), rs.randn(M - M // 2, d) * 3])
        A = rs.randn(d, d)
        cov = 0.2 * (A @ A.T + d * np.eye(d)) / d
        w = rs.uniform(0.1, 2.0, N)
        if d == 1:
            means_in, x_in, cov_in = means[:, 0], x[:, 0], float(cov[0, 0])
        else:
            means_in, x_in, cov_in = means, x, cov
        out['x_%d' % k], out['means_%d' % k], out['cov_%d' % k], out['w_%d' % k] = x_in, means_in, np.asarray(cov_in), w
        out['pdf_%d' % k] = GMDistribution.pdf(x_in, means_in, cov=cov_in, weights=w)
        out['logpdf_%d' % k] = GMDistribution.logpdf(x_in, means_in, cov=cov_in, weights=w)
        out['pdf_now_%d' % k] = GMDistribution.pdf(x_in, means_in, cov=cov_in)      # weights=None
        cases.append(k)
    out['pdf_single'] = GMDistribution.pdf(out['x_1'][3], out['means_1'], cov=out['cov_1'], weights=out['w_1'])
    out['cases'] = np.array(cases)
    np.savez_compressed(os.path.join(GOLDEN, 'gm_pdf.npz'), **out)
    print('gm_pdf:', len(cases), 'cases')


def make_weighted(elfi):
    """weighted_var / weighted_sample_quantile of the real reference (elfi/methods/utils.py:108-139, 379-411)."""
    from elfi.methods.utils import weighted_sample_quantile, weighted_var
    rs = np.random.RandomState(4242)
    out, cases = {}, []
    for k, (n, m, have_w) in enumerate([(7, 1, True), (500, 2, True), (500, 2, False), (6000, 5, True), (800, 64, True),
                                        (257, 300, True)]):
        x = rs.randn(n, m) * rs.uniform(0.1, 30, m) + rs.uniform(-5, 5, m)
        w = rs.uniform(0.0, 3.0, n) ** 2 if have_w else None
        xin = x[:, 0] if m == 1 else x
        out['x_%d' % k] = xin
        if have_w:
            out['w_%d' % k] = w
        out['var_%d' % k] = np.asarray(weighted_var(xin, w))
        cases.append(k)
    out['var_cases'] = np.array(cases)
    # quantiles: random weights, equal weights with alpha on the boundaries k / n, ties in x, alpha = 0 and 1
    qcases = []
    for k, (n, equal, ties) in enumerate([(1000, False, False), (64, True, False), (64, True, True), (5, False, True),
                                           (20000, False, False)]):
        x = rs.randn(n)
        if ties:
            x = np.round(x, 1)
        w = None if equal else rs.uniform(0.01, 1.0, n)
        alphas = np.array([0.0, 1e-9, 0.01, 0.25, 0.5, 0.75, 0.999, 1.0] + ([j / n for j in (1, 2, n // 2, n - 1)] if equal else []))
        out['qx_%d' % k] = x
        if w is not None:
            out['qw_%d' % k] = w
        out['qalpha_%d' % k] = alphas
        out['q_%d' % k] = np.array([weighted_sample_quantile(x, a, weights=w) for a in alphas])
        qcases.append(k)
    out['q_cases'] = np.array(qcases)
    np.savez_compressed(os.path.join(GOLDEN, 'weighted_stats.npz'), **out)
    print('weighted_stats:', len(cases), 'variance cases,', len(qcases), 'quantile cases')


def main(argv):
    os.makedirs(GOLDEN, exist_ok=True)
    elfi = ref_shim.install()
    which = set(argv) or {'ma2', 'gauss', 'adaptive', 'metrics', 'gp', 'gm', 'posterior', 'weighted', 'docrun'}
    if 'ma2' in which:
        make_ma2(elfi)
    if 'adaptive' in which:
        make_adaptive(elfi)
    if 'metrics' in which:
        make_metrics(elfi)
    if 'gm' in which:
        make_gm(elfi)
    if 'gauss' in which:
        make_gauss(elfi)
    if 'weighted' in which:
        make_weighted(elfi)
    if 'gp' in which:
        try:
            import make_golden_gp
        except ImportError:
            print('gp fixtures: generator not present yet')
        else:
            make_golden_gp.main(elfi, GOLDEN)
    if 'docrun' in which:          # ~1 min: the documented BOLFI run replayed through the reference's loop
        import make_golden_gp
        make_golden_gp.make_doc_run(elfi, GOLDEN)
    if 'posterior' in which:
        import make_golden_posterior
        make_golden_posterior.main(elfi, GOLDEN)


if __name__ == '__main__':
    main(sys.argv[1:])
