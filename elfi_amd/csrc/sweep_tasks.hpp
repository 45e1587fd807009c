// Task list of the resident sweep kernel (gp_fit.hip: sweep_kernel).  Pure C++ (no HIP): tests/test_sweep_tasks.py
// compiles it with g++ and checks the invariants the kernel relies on.
//
// Row blocks: 0..nb-1 the 128-row blocks of A, nb the y block, nb+1+r the L^-T row r.
//   type 0  potf2(k)       diagonal block k                          rb = c = k
//   type 1  trsm(rb, k)    row block rb of panel k times W11_k^T     c unused (0)
//   type 2  upd(rb, c, k)  C(rb, c) -= P(rb, k) P(c, k)^T            c = block column (an A row block as operand)
// Order: a topological order of the dependencies (a task needs: potf2 -- k updates on its tile; trsm -- potf2(k) and
// all earlier updates of its tile; upd -- the solves of both operands (potf2(k) for the L^-T row k, whose "panel" is
// the diagonal block of L^-T itself) and the earlier updates of its tile) with the next panel's critical tasks first:
//   block k:  potf2(k) | trsm(k+1, k), upd(k+1, k+1, k) | rest of panel k-1 (columns >= k+2) |
//             the other solves of panel k and its updates of columns k+1 and k+2
#pragma once

#include <vector>

namespace elfihip {

struct SweepTask {
  int type, rb, c, k;
};

inline void sweep_build_tasks(int nb, std::vector<SweepTask>* out) {
  std::vector<SweepTask>& L = *out;
  L.clear();
  const int Y = nb;
  auto wt = [&](int r) { return nb + 1 + r; };
  auto emit_near = [&](int k) {  // the other solves of panel k, then its updates of block columns k+1 and k+2
    for (int i = k + 2; i < nb; ++i) L.push_back({1, i, 0, k});
    L.push_back({1, Y, 0, k});
    for (int r = 0; r < k; ++r) L.push_back({1, wt(r), 0, k});
    for (int c = k + 1; c <= k + 2 && c < nb; ++c) {
      for (int i = c; i < nb; ++i)
        if (!(i == k + 1 && c == k + 1)) L.push_back({2, i, c, k});
      L.push_back({2, Y, c, k});
      for (int r = 0; r <= k; ++r) L.push_back({2, wt(r), c, k});
    }
  };
  auto emit_far = [&](int k) {  // the rest of panel k: block columns >= k+3
    for (int c = k + 3; c < nb; ++c) {
      for (int i = c; i < nb; ++i) L.push_back({2, i, c, k});
      L.push_back({2, Y, c, k});
      for (int r = 0; r <= k; ++r) L.push_back({2, wt(r), c, k});
    }
  };
  for (int k = 0; k < nb; ++k) {
    L.push_back({0, k, k, k});
    if (k + 1 < nb) {
      L.push_back({1, k + 1, 0, k});
      L.push_back({2, k + 1, k + 1, k});
    }
    if (k > 0) emit_far(k - 1);
    emit_near(k);
  }
}

}  // namespace elfihip
