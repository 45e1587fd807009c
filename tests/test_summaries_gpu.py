"""GPU: summary statistics and the fused MA2 path vs NumPy / the oracle -- BIT-EXACT.

The kernels follow NumPy's pairwise summation order (csrc/summaries.hip), so equality is exact,
including rows longer than NumPy's 128-element pairwise block and odd / unaligned shapes.
"""
import numpy as np
import pytest

import distance_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('n,L', [(1, 100), (7, 3), (1000, 100), (999, 50), (513, 99), (300, 128), (300, 129),
                                 (200, 130), (64, 1000), (50, 1003), (20000, 100)])
def test_row_summaries_bit_exact(hip_ctx, n, L):
    import elfi_amd
    rs = np.random.RandomState(n + L)
    x = rs.randn(n, L) * rs.uniform(0.1, 50, L) + rs.uniform(-3, 3, L)
    assert np.array_equal(elfi_amd.ss_mean(x), O.ss_mean(x))
    assert np.array_equal(elfi_amd.ss_var(x), O.ss_var(x))
    for lag in (1, 2, min(7, L - 1)):
        if lag < L:
            assert np.array_equal(elfi_amd.autocov(x, lag), O.autocov(x, lag)), (n, L, lag)


def test_summary_on_observed_data_and_views(hip_ctx):
    import elfi_amd
    rs = np.random.RandomState(1)
    y_obs = rs.randn(1, 100)                    # ELFI calls summaries on the observed data too (leading dim 1)
    assert np.array_equal(elfi_amd.autocov(y_obs), O.autocov(y_obs))
    assert np.array_equal(elfi_amd.autocov(y_obs[0], 2), O.autocov(y_obs[0], 2))   # 1-d input -> atleast_2d
    big = rs.randn(300, 120)
    view = big[:, 10:110]                        # row pitch 120, width 100
    assert np.array_equal(elfi_amd.autocov(view, 1), O.autocov(view, 1))
    assert np.array_equal(elfi_amd.ss_var(big[::2]), O.ss_var(big[::2]))
    assert np.array_equal(elfi_amd.ss_mean(big.astype(np.float32)), O.ss_mean(big.astype(np.float32).astype(np.float64)))
    with pytest.raises(ValueError):
        elfi_amd.autocov(big, 0)
    with pytest.raises(ValueError):
        elfi_amd.autocov(big, 120)
    assert elfi_amd.ss_mean(np.empty((0, 5))).shape == (0,)


@pytest.mark.parametrize('batch,n_obs', [(1, 100), (1000, 100), (4097, 100), (333, 37), (200, 200)])
def test_fused_ma2_path_bit_exact(hip_ctx, batch, n_obs):
    """MA2 simulator arithmetic + autocov(1), autocov(2) + euclidean distance in one kernel ==
    the reference chain MA2 -> autocov -> cdist on the same MT19937 draws (elfi/examples/ma2.py)."""
    import elfi_amd
    rs = np.random.RandomState(batch)
    t1 = rs.uniform(-2, 2, batch)
    t2 = rs.uniform(-1, 1, batch)
    seed = 20170530
    x = O.MA2(t1, t2, n_obs=n_obs, batch_size=batch, random_state=np.random.RandomState(seed))
    y_obs = O.MA2(0.6, 0.2, n_obs=n_obs, random_state=np.random.RandomState(1))
    obs = (O.autocov(y_obs), O.autocov(y_obs, 2))
    S1, S2 = O.autocov(x), O.autocov(x, 2)
    d = O.make_distance('euclidean')(S1, S2, observed=obs)
    w = np.random.RandomState(seed).randn(batch, n_obs + 2)     # the draw MA2 makes internally
    g1, g2, gd = elfi_amd.ma2_distance(w, t1, t2, np.concatenate(obs))
    assert np.array_equal(g1, S1) and np.array_equal(g2, S2) and np.array_equal(gd, d)
    # the generic summaries give the same on the materialised x
    assert np.array_equal(elfi_amd.autocov(x), S1) and np.array_equal(elfi_amd.autocov(x, 2), S2)


def test_ma2_tutorial_golden_through_summaries_and_distance(hip_ctx, golden_dir):
    """The documented MA2 run (docs/usage/tutorial.rst:396): recompute every batch's distance from the
    reference's recorded summaries and reproduce the threshold 0.116859716394976."""
    import os
    import elfi_amd
    g = np.load(os.path.join(golden_dir, 'ma2_tutorial.npz'))
    obs = g['observed']
    assert np.array_equal(elfi_amd.autocov(g['y_obs']), obs[0, :1])
    assert np.array_equal(elfi_amd.autocov(g['y_obs'], 2), obs[0, 1:])
    op = elfi_amd.HipDiscrepancy('euclidean')
    d = np.concatenate([op(s1, s2, observed=(obs[:, 0], obs[:, 1])) for s1, s2 in zip(g['S1'], g['S2'])])
    assert np.array_equal(d[:len(g['d0'])], g['d0'])
    thr = np.sort(d)[999]
    assert repr(float(thr)) == '0.116859716394976'


@pytest.mark.parametrize('n,n_obs', [(1, 100), (777, 100), (50000, 100), (4096, 37), (1000, 126)])
def test_ma2_with_the_noise_drawn_in_the_kernel(hip_ctx, n, n_obs):
    """elfihip_ma2_draw_distance_dev (synthetic-throughput form of examples/ma2.py:11-59): the white noise never exists
    in memory, yet S1, S2 and the distance equal -- bit for bit -- the fused path run on the matrix elfihip_randn_dev
    writes for the same (seed, stream), which in turn is bit-identical to NumPy on those values
    (test_fused_ma2_path_bit_exact)."""
    import ctypes as C
    import torch
    lib = hip_ctx.lib
    L = n_obs + 2
    rs = np.random.RandomState(n)
    t1 = torch.from_numpy(rs.uniform(-1, 1, n)).cuda()
    t2 = torch.from_numpy(rs.uniform(-0.5, 0.5, n)).cuda()
    W = torch.empty(n, L, dtype=torch.float64, device='cuda')
    out = [torch.empty(n, dtype=torch.float64, device='cuda') for _ in range(6)]
    torch.cuda.synchronize()
    seed, stream = C.c_uint64(2024), C.c_uint64(5)
    assert lib.elfihip_randn_dev(hip_ctx.handle, seed, stream, n * L, C.c_double(0.0), C.c_double(1.0), W.data_ptr()) == 0
    assert lib.elfihip_ma2_distance_dev(hip_ctx.handle, W.data_ptr(), n, n_obs, L, t1.data_ptr(), t2.data_ptr(),
                                        C.c_double(0.3), C.c_double(0.1), out[0].data_ptr(), out[1].data_ptr(),
                                        out[2].data_ptr()) == 0
    assert lib.elfihip_ma2_draw_distance_dev(hip_ctx.handle, seed, stream, n, n_obs, t1.data_ptr(), t2.data_ptr(),
                                             C.c_double(0.3), C.c_double(0.1), out[3].data_ptr(), out[4].data_ptr(),
                                             out[5].data_ptr()) == 0
    hip_ctx.synchronize()
    for a, b in zip(out[:3], out[3:]):
        assert np.array_equal(a.cpu().numpy(), b.cpu().numpy())
    # and against NumPy on the drawn noise
    w = W.cpu().numpy()
    x = w[:, 2:] + t1.cpu().numpy()[:, None] * w[:, 1:-1] + t2.cpu().numpy()[:, None] * w[:, :-2]
    s1 = np.mean(x[:, 1:] * x[:, :-1], axis=1)
    assert np.array_equal(out[3].cpu().numpy(), s1)
