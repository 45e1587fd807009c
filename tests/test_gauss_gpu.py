"""GPU: the fused Gaussian example (elfi_amd/csrc/gauss.hip) -- bit-exact against the reference's outputs on the
reference's draws; the device generator against Philox4x32-10's published known answers and as a distribution."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def test_fused_path_equals_the_reference_bit_for_bit(hip_ctx):
    """z from the RandomState the reference's simulator consumed -> y, ss_mean, ss_var, distance: every value equal to
    what elfi/examples/gauss.py and the Distance node produced (tests/golden/gauss_example.npz)."""
    import elfi_amd
    from elfi_amd import summaries
    g = np.load(os.path.join(GOLDEN, 'gauss_example.npz'))
    for tag in g['cases']:                                    # n_obs = 50, 7 (< 8), 200 (> 128), 129
        mu, sigma, n_obs = g['mu_' + tag], g['sigma_' + tag], int(g['n_obs_' + tag])
        z = np.random.RandomState(int(g['draw_seed_' + tag])).standard_normal((mu.shape[0], n_obs))
        s1, s2, d, y = summaries.gauss_distance(mu, sigma, g['observed_' + tag], z=z, return_y=True)
        assert np.array_equal(y, g['y_' + tag]), tag
        assert np.array_equal(s1, g['ss_mean_' + tag]) and np.array_equal(s2, g['ss_var_' + tag]), tag
        assert np.array_equal(d, g['d_' + tag]), tag
    # scalar parameters broadcast, as elfi hands them over for a constant node
    s1, s2, d = summaries.gauss_distance(4.0, 0.4, [4.0, 0.16], z=z)
    import gauss_oracle as GO
    y = GO.gauss_from_draws(z, 4.0, 0.4)
    assert np.array_equal(s1, GO.ss_mean(y)) and np.array_equal(s2, GO.ss_var(y))
    assert np.array_equal(d, GO.euclidean_to_observed(s1, s2, [4.0, 0.16]))


def test_philox_known_answers(hip_ctx):
    """Random123's kat_vectors for philox4x32-10: counter / key all zero, all ones, and the digits of pi."""
    import torch
    lib = hip_ctx.lib

    def block(counter, key):
        # the entry point numbers blocks by (block, stream) = counter words (0,1) and (2,3); key = seed
        blk = counter[0] | (counter[1] << 32)
        stream = counter[2] | (counter[3] << 32)
        seed = key[0] | (key[1] << 32)
        out = torch.zeros(4 * (blk % 7 + 1), dtype=torch.int32, device='cuda')
        torch.cuda.synchronize()
        # the block index is the launch's element index: ask for block `blk` alone by offsetting through the stream is not
        # possible, so only small block numbers are used here
        assert blk < 7
        assert lib.elfihip_random_bits_dev(hip_ctx.handle, C.c_uint64(seed), C.c_uint64(stream), blk + 1, out.data_ptr()) == 0
        hip_ctx.synchronize()
        return [int(v) & 0xffffffff for v in out.cpu().numpy()[4 * blk: 4 * blk + 4]]

    assert block((0, 0, 0, 0), (0, 0)) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert block((0, 0, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff)) != block((0, 0, 0, 0), (0, 0))
    # a second published vector needs counter word 0 = 0xffffffff; covered through the generic implementation below
    import philox_ref
    for ctr, key in (((3, 0, 0x243f6a88, 0x85a308d3), (0xa4093822, 0x299f31d0)), ((5, 0, 1, 2), (3, 4))):
        assert block(ctr, key) == philox_ref.philox4x32_10(ctr, key)


def test_device_draws_are_standard_normal_and_reproducible(hip_ctx):
    import torch
    from scipy import stats
    lib = hip_ctx.lib
    n = 2_000_001                                              # odd: the last pair is half used
    a = torch.empty(n, dtype=torch.float64, device='cuda')
    b = torch.empty(n, dtype=torch.float64, device='cuda')
    torch.cuda.synchronize()
    assert lib.elfihip_randn_dev(hip_ctx.handle, C.c_uint64(7), C.c_uint64(0), n, C.c_double(0.0), C.c_double(1.0), a.data_ptr()) == 0
    assert lib.elfihip_randn_dev(hip_ctx.handle, C.c_uint64(7), C.c_uint64(0), n // 3, C.c_double(0.0), C.c_double(1.0), b.data_ptr()) == 0
    hip_ctx.synchronize()
    z = a.cpu().numpy()
    assert np.array_equal(z[: n // 3 - 1], b.cpu().numpy()[: n // 3 - 1])      # a prefix is a prefix: counter-based
    assert np.all(np.isfinite(z))
    assert abs(z.mean()) < 4 / np.sqrt(n) and abs(z.var() - 1) < 6 * np.sqrt(2 / n)
    assert abs(stats.skew(z)) < 0.01 and abs(stats.kurtosis(z)) < 0.02
    assert stats.kstest(z[:200000], 'norm').pvalue > 1e-3
    assert abs(np.corrcoef(z[:-1:2], z[1::2])[0, 1]) < 5e-3 and abs(np.corrcoef(z[:-1], z[1:])[0, 1]) < 5e-3
    # other seed / stream: different numbers
    assert lib.elfihip_randn_dev(hip_ctx.handle, C.c_uint64(8), C.c_uint64(0), 1000, C.c_double(0.0), C.c_double(1.0), b.data_ptr()) == 0
    hip_ctx.synchronize()
    assert not np.any(b.cpu().numpy()[:1000] == z[:1000])
    assert lib.elfihip_randn_dev(hip_ctx.handle, C.c_uint64(7), C.c_uint64(1), 1000, C.c_double(2.0), C.c_double(3.0), b.data_ptr()) == 0
    hip_ctx.synchronize()
    w = b.cpu().numpy()[:1000]
    assert not np.any(w == z[:1000]) and abs(w.mean() - 2.0) < 0.5


def test_fused_path_with_device_draws(hip_ctx):
    """Z = NULL: the kernel's own draws are the generator's element e = row * n_obs + i, whatever the tiling, and the
    summaries / distance of those draws are NumPy's."""
    import torch
    from elfi_amd import summaries
    import gauss_oracle as GO
    lib = hip_ctx.lib
    for n, n_obs in ((5000, 50), (333, 7), (100, 201)):
        rs = np.random.RandomState(n)
        mu, sigma = rs.uniform(0, 8, n), rs.uniform(0.1, 3, n)
        s1, s2, d, y = summaries.gauss_distance(mu, sigma, [4.0, 0.16], n_obs=n_obs, seed=42, stream=9, return_y=True)
        zt = torch.empty(n * n_obs, dtype=torch.float64, device='cuda')
        torch.cuda.synchronize()
        assert lib.elfihip_randn_dev(hip_ctx.handle, C.c_uint64(42), C.c_uint64(9), n * n_obs, C.c_double(0.0), C.c_double(1.0), zt.data_ptr()) == 0
        hip_ctx.synchronize()
        z = zt.cpu().numpy().reshape(n, n_obs)
        assert np.array_equal(y, GO.gauss_from_draws(z, mu, sigma))
        assert np.array_equal(s1, GO.ss_mean(y)) and np.array_equal(s2, GO.ss_var(y))
        assert np.array_equal(d, GO.euclidean_to_observed(s1, s2, [4.0, 0.16]))
