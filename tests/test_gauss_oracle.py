"""The Gaussian example's oracle (oracle/gauss_oracle.py) against what the reference's own functions produced
(tests/golden/gauss_example.npz, oracle/make_golden.py make_gauss): bit for bit."""
import os

import numpy as np

from conftest import GOLDEN
import gauss_oracle as GO


def test_oracle_reproduces_the_reference_functions():
    g = np.load(os.path.join(GOLDEN, 'gauss_example.npz'))
    for tag in g['cases']:
        mu, sigma, n_obs = g['mu_' + tag], g['sigma_' + tag], int(g['n_obs_' + tag])
        z = GO.draws(np.random.RandomState(int(g['draw_seed_' + tag])), mu.shape[0], n_obs)
        y = GO.gauss_from_draws(z, mu, sigma)
        assert np.array_equal(y, g['y_' + tag]), tag                      # ss.norm.rvs == standard_normal * scale + loc
        assert np.array_equal(GO.ss_mean(y), g['ss_mean_' + tag])
        assert np.array_equal(GO.ss_var(y), g['ss_var_' + tag])
        d = GO.euclidean_to_observed(g['ss_mean_' + tag], g['ss_var_' + tag], g['observed_' + tag])
        assert np.array_equal(d, g['d_' + tag]), tag


def test_oracle_reproduces_a_batch_of_the_example_model():
    g = np.load(os.path.join(GOLDEN, 'gauss_example.npz'))
    y = g['model_gauss']
    assert np.array_equal(GO.ss_mean(y), g['model_ss_mean']) and np.array_equal(GO.ss_var(y), g['model_ss_var'])
    d = GO.euclidean_to_observed(g['model_ss_mean'], g['model_ss_var'], g['model_observed'])
    assert np.array_equal(d, g['model_d'])
