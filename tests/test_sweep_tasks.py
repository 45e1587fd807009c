"""The static task list of the resident sweep kernel (elfi_amd/csrc/sweep_tasks.hpp, used by gp_fit.hip's sweep_kernel
under ELFIHIP_SWEEP=1), checked on the CPU for panel groups G = 1, 2, 3, 4:

  * topological: executed one by one in list order, every task finds what it waits for (diagonal blocks done, panels
    solved for its row blocks, earlier updates of its tile) -- so a workgroup that has drawn a task only ever waits for
    tasks drawn before it, which is the kernel's no-deadlock argument;
  * it computes the factorisation: the list is EXECUTED with NumPy on small blocks (the same three task bodies as the
    kernel) and gives L, L^-T and z = L^-1 y of a random SPD matrix;
  * look-ahead: the next diagonal block follows directly the two tasks it waits for.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FIELDS = ('type', 'rb', 'c', 'k0', 'kun', 'prior', 'need_pdone', 'need_rb', 'need_c', 'beta0')


@pytest.fixture(scope='module')
def lib(tmp_path_factory):
    out = str(tmp_path_factory.mktemp('sweep') / 'libsweep_tasks_test.so')
    subprocess.check_call(['g++', '-O2', '-std=c++17', '-shared', '-fPIC', '-o', out,
                           os.path.join(HERE, 'cpp', 'sweep_tasks_capi.cpp')])
    lb = C.CDLL(out)
    lb.sweep_tasks.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int)]
    lb.sweep_tasks.restype = C.c_int
    return lb


def tasks(lib, nb, G):
    n = lib.sweep_tasks(nb, G, None)
    buf = np.empty((n, len(FIELDS)), dtype=np.int32)
    assert lib.sweep_tasks(nb, G, buf.ctypes.data_as(C.POINTER(C.c_int))) == n
    return [dict(zip(FIELDS, (int(v) for v in row))) for row in buf]


def run_list(L, nb, b, K, y):
    """Execute the task list on b x b blocks exactly as sweep_kernel does (same operands, same in-place rules) and
    check every wait condition on the way."""
    n = nb * b
    A = np.zeros(((nb + 1) * b, n))
    A[:n] = np.tril(K)
    A[n] = y                                       # the y block: first row y^T, the rest zero
    WT = np.triu(np.random.RandomState(7).randn(n, n) * 1e3)   # stale contents of an earlier factorisation where the
    for k in range(nb):                                       # kernel overwrites (beta0); zero left of the diagonal blocks
        WT[k * b:(k + 1) * b, :k * b] = 0.0
    w11 = [None] * nb
    pdone, solved, cnt = 0, np.zeros(2 * nb + 1, dtype=int), np.zeros((2 * nb + 1, nb), dtype=int)

    def rows(rb):
        return (A, rb * b) if rb <= nb else (WT, (rb - nb - 1) * b)

    for pos, t in enumerate(L):
        tile_col = t['c'] if t['type'] == 2 else t['k0']
        assert pdone >= t['need_pdone'] and solved[t['rb']] >= t['need_rb'], (pos, t)
        assert t['type'] != 2 or solved[t['c']] >= t['need_c'], (pos, t)
        assert cnt[t['rb'], tile_col] == t['prior'], (pos, t)
        k0, ks = t['k0'] * b, slice(t['k0'] * b, (t['k0'] + t['kun']) * b)
        if t['type'] == 0:
            k = t['k0']
            Lkk = np.linalg.cholesky(np.tril(A[k0:k0 + b, k0:k0 + b]) + np.tril(A[k0:k0 + b, k0:k0 + b], -1).T)
            A[k0:k0 + b, k0:k0 + b] = Lkk
            w11[k] = np.linalg.inv(Lkk)
            WT[k0:k0 + b, k0:k0 + b] = w11[k].T
            pdone = k + 1
        elif t['type'] == 1:
            M, r0 = rows(t['rb'])
            M[r0:r0 + b, k0:k0 + b] = M[r0:r0 + b, k0:k0 + b] @ w11[t['k0']].T
            solved[t['rb']] = t['k0'] + 1
        else:
            M, r0 = rows(t['rb'])
            c0 = t['c'] * b
            prod = M[r0:r0 + b, ks] @ A[c0:c0 + b, ks].T
            M[r0:r0 + b, c0:c0 + b] = -prod if t['beta0'] else M[r0:r0 + b, c0:c0 + b] - prod
            cnt[t['rb'], t['c']] += 1
    return A, WT


@pytest.mark.parametrize('G', [1, 2, 3, 4])
@pytest.mark.parametrize('nb', [1, 2, 3, 5, 8, 13])
def test_list_is_topological_and_computes_the_factorisation(lib, nb, G):
    b = 3
    n = nb * b
    rs = np.random.RandomState(100 * nb + G)
    Q = rs.randn(n, n)
    K = Q @ Q.T + n * np.eye(n)
    y = rs.randn(n)
    L = tasks(lib, nb, G)
    assert len({tuple(t.values()) for t in L}) == len(L)                  # no task twice
    A, WT = run_list(L, nb, b, K, y)
    Lref = np.linalg.cholesky(K)
    np.testing.assert_allclose(np.tril(A[:n]), Lref, rtol=0, atol=1e-10 * np.abs(Lref).max())
    np.testing.assert_allclose(np.triu(WT), np.linalg.inv(Lref).T, rtol=0, atol=1e-10)
    assert np.all(np.tril(WT, -1) == 0)
    np.testing.assert_allclose(A[n], np.linalg.solve(Lref, y), rtol=0, atol=1e-10)


@pytest.mark.parametrize('nb,G', [(32, 1), (32, 4), (40, 4), (33, 2)])
def test_large_lists_are_topological_and_the_next_diagonal_block_comes_first(lib, nb, G):
    L = tasks(lib, nb, G)
    pdone, solved, cnt = 0, np.zeros(2 * nb + 1, dtype=int), np.zeros((2 * nb + 1, nb), dtype=int)
    for pos, t in enumerate(L):
        col = t['c'] if t['type'] == 2 else t['k0']
        assert pdone >= t['need_pdone'] and solved[t['rb']] >= t['need_rb'] and cnt[t['rb'], col] == t['prior'], (pos, t)
        assert t['type'] != 2 or solved[t['c']] >= t['need_c'], (pos, t)
        if t['type'] == 0:
            pdone = t['k0'] + 1
        elif t['type'] == 1:
            solved[t['rb']] = t['k0'] + 1
        else:
            cnt[t['rb'], t['c']] += 1
    assert pdone == nb
    where = {(t['type'], t['rb'], t['k0']): i for i, t in enumerate(L) if t['type'] != 2}
    for k in range(nb - 1):
        p = where[(0, k, k)]
        assert (L[p + 1]['type'], L[p + 1]['rb'], L[p + 1]['k0']) == (1, k + 1, k)
        assert (L[p + 2]['type'], L[p + 2]['rb'], L[p + 2]['c']) == (2, k + 1, k + 1)
    n_upd = sum(1 for t in L if t['type'] == 2)
    n_upd_plain = sum(1 for t in tasks(lib, nb, 1) if t['type'] == 2)
    assert n_upd <= n_upd_plain and (G == 1 or n_upd < 0.6 * n_upd_plain)   # grouping cuts the number of updates
