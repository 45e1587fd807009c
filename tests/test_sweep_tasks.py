"""The static task list of the resident sweep kernel (elfi_amd/csrc/sweep_tasks.hpp, used by gp_fit.hip's sweep_kernel
under ELFIHIP_SWEEP=1): the invariants the kernel's no-deadlock argument and its results rest on, checked on the CPU.

  * topological: executed one by one in list order, every task finds its inputs ready (so a workgroup that has drawn a
    task only ever waits for tasks drawn before it);
  * complete: every diagonal block is factored once, every (row block, panel) is solved once, every tile receives exactly
    the updates of the right-looking algorithm (A / y tile (i, c): panels 0..c-1; L^-T tile (r, c): panels r..c-1), in
    ascending panel order;
  * look-ahead: potf2(k+1) comes right after the two tasks it waits for, before the bulk of panel k.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def lib(tmp_path_factory):
    out = str(tmp_path_factory.mktemp('sweep') / 'libsweep_tasks_test.so')
    subprocess.check_call(['g++', '-O2', '-std=c++17', '-shared', '-fPIC', '-o', out,
                           os.path.join(HERE, 'cpp', 'sweep_tasks_capi.cpp')])
    lb = C.CDLL(out)
    lb.sweep_tasks.argtypes = [C.c_int, C.POINTER(C.c_int)]
    lb.sweep_tasks.restype = C.c_int
    return lb


def tasks(lib, nb):
    n = lib.sweep_tasks(nb, None)
    buf = np.empty((n, 4), dtype=np.int32)
    assert lib.sweep_tasks(nb, buf.ctypes.data_as(C.POINTER(C.c_int))) == n
    return buf


@pytest.mark.parametrize('nb', [1, 2, 3, 4, 5, 8, 13, 32, 40])
def test_list_is_topological_and_complete(lib, nb):
    L = tasks(lib, nb)
    Y = nb
    pdone = 0
    solved = np.zeros(2 * nb + 1, dtype=int)       # panels solved per row block
    cnt = np.zeros((2 * nb + 1, nb), dtype=int)    # updates applied per tile
    seen = set()
    for pos, (typ, rb, c, k) in enumerate(L):
        key = (int(typ), int(rb), int(c), int(k))
        assert key not in seen, 'duplicate task %r' % (key,)
        seen.add(key)
        r_wt = rb - nb - 1
        prior = k - r_wt if r_wt >= 0 else k
        if typ == 0:
            assert rb == k == c and pdone == k and cnt[k, k] == k, (pos, key)
            pdone = k + 1
        elif typ == 1:
            assert (r_wt < k) if r_wt >= 0 else (rb > k), (pos, key)      # rows below, the y block, L^-T rows above
            assert pdone >= k + 1 and cnt[rb, k] == prior and solved[rb] <= k, (pos, key)
            solved[rb] = k + 1
        else:
            assert k < c < nb, (pos, key)
            first = pdone >= k + 1 if r_wt == k else solved[rb] >= k + 1
            assert first and solved[c] >= k + 1 and cnt[rb, c] == prior, (pos, key)
            cnt[rb, c] += 1
    assert pdone == nb
    for i in range(nb):                                   # rows of A: tile (i, c <= i) got panels 0..c-1, solved c < i
        for c in range(i + 1):
            assert cnt[i, c] == c
        assert solved[i] == i                              # panels 0..i-1 (panel i is the diagonal block itself)
    assert all(cnt[Y, c] == c for c in range(nb)) and solved[Y] == nb
    for r in range(nb):                                   # L^-T row r: tile (r, c > r) got panels r..c-1
        for c in range(r + 1, nb):
            assert cnt[nb + 1 + r, c] == c - r
        assert solved[nb + 1 + r] == (nb if r < nb - 1 else 0)


@pytest.mark.parametrize('nb', [4, 32])
def test_next_diagonal_block_is_drawn_before_the_bulk(lib, nb):
    L = [tuple(int(v) for v in t) for t in tasks(lib, nb)]
    for k in range(nb - 1):
        p = L.index((0, k, k, k))
        assert L[p + 1] == (1, k + 1, 0, k) and L[p + 2] == (2, k + 1, k + 1, k)
        q = L.index((0, k + 1, k + 1, k + 1))
        far = [i for i, t in enumerate(L) if t[0] == 2 and t[3] == k and t[2] >= k + 3]
        assert all(i > q for i in far)        # the bulk of panel k comes after potf2(k+1) has been drawn
    assert sum(1 for t in L if t[0] == 1) == nb * nb      # per panel: nb-1-k rows below + y + k rows of L^-T = nb
