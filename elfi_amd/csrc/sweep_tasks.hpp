// Task list of the resident sweep kernel (gp_fit.hip: sweep_kernel).  Pure C++ (no HIP): tests/test_sweep_tasks.py
// compiles it with g++, checks the invariants the kernel relies on and EXECUTES the list with NumPy on small blocks.
//
// Row blocks: 0..nb-1 the 128-row blocks of A, nb the y block, nb+1+r the L^-T row r (its tiles left of the diagonal
// stay zero, so an update that spans panels before r simply multiplies zeros there).
//   type 0  potf2(k)              diagonal block k                                           rb = c = k0 = k
//   type 1  trsm(rb, k)           row block rb of panel k times W11_k^T, in place            k0 = k
//   type 2  upd(rb, c, k0, kun)   C(rb, c) -= P(rb, k0 .. k0+kun) P(c, k0 .. k0+kun)^T       beta0: C = - ... (first touch
//                                                                                            of an L^-T tile)
// Panels are grouped (G per group, the last group may be shorter).  A tile right of its group receives the group's G
// panels in ONE update once the group is solved (depth 128 G: the read-modify-write traffic on the tile and the number of
// tasks drop by G); a tile whose column lies inside the group receives the group's earlier panels in one update just
// before its own panel (left-looking inside the group).  G = 1 is the plain right-looking sweep.
//
// Every task carries what it has to wait for: `need_pdone` diagonal blocks done, `need_rb` / `need_c` panels solved for
// its row block / its column's row block, `prior` earlier updates on its tile (the kernel counts updates per tile).
// Order: a topological order of these dependencies with the next panel's two critical tasks first:
//   block k:  potf2(k) | trsm(k+1, k), update of tile (k+1, k+1) | far updates left over from the group that ended at
//             k-1 (columns >= k+2) | the other solves of panel k, the updates of column k+1 and, at a group's end, k+2
#pragma once

#include <vector>

namespace elfihip {

struct SweepTask {
  int type, rb, c, k0, kun, prior, need_pdone, need_rb, need_c, beta0;
};

inline void sweep_build_tasks(int nb, int G, std::vector<SweepTask>* out) {
  std::vector<SweepTask>& L = *out;
  L.clear();
  if (G < 1) G = 1;
  const int Y = nb, nrb = 2 * nb + 1;
  std::vector<int> cnt((size_t)nrb * nb, 0);  // updates emitted so far per tile = `prior` of the next one
  auto wt = [&](int r) { return nb + 1 + r; };
  auto potf2 = [&](int k) { L.push_back({0, k, k, k, 1, cnt[(size_t)k * nb + k], 0, 0, 0, 0}); };
  auto trsm = [&](int rb, int k) { L.push_back({1, rb, 0, k, 1, cnt[(size_t)rb * nb + k], k + 1, 0, 0, 0}); };
  auto upd = [&](int rb, int c, int k0, int kun) {
    const int r = rb - nb - 1;  // L^-T row or negative
    SweepTask t{2, rb, c, k0, kun, cnt[(size_t)rb * nb + c], 0, 0, k0 + kun, 0};
    if (r < 0) {
      t.need_rb = k0 + kun;
    } else {
      if (r >= k0) {  // the range contains the row's own diagonal block: first touch of this tile
        t.need_pdone = r + 1;
        t.beta0 = 1;
      }
      if (k0 + kun - 1 > r) t.need_rb = k0 + kun;  // panels after r in the range are solved tiles of this row
    }
    L.push_back(t);
    ++cnt[(size_t)rb * nb + c];
  };
  // every row block that has a tile in block column c and takes part in an update by panels [k0, k0 + kun)
  auto column = [&](int c, int k0, int kun, bool skip_diag) {
    for (int i = c; i < nb; ++i)
      if (!(skip_diag && i == c)) upd(i, c, k0, kun);
    upd(Y, c, k0, kun);
    for (int r = 0; r < k0 + kun && r < c; ++r) upd(wt(r), c, k0, kun);
  };
  int far_g0 = -1, far_kun = 0, far_from = 0;  // far updates of the last finished group still to be emitted
  for (int k = 0; k < nb; ++k) {
    const int g0 = (k / G) * G;
    const int gend = g0 + G < nb ? g0 + G : nb;  // first panel after this group
    const bool last = (k == gend - 1);           // k ends its group
    potf2(k);
    if (k + 1 < nb) {
      trsm(k + 1, k);
      if (last)
        upd(k + 1, k + 1, g0, gend - g0);        // the group's one update of the next diagonal tile
      else
        upd(k + 1, k + 1, g0, k + 1 - g0);       // left-looking inside the group
    }
    if (far_g0 >= 0) {
      for (int c = far_from; c < nb; ++c) column(c, far_g0, far_kun, false);
      far_g0 = -1;
    }
    for (int i = k + 2; i < nb; ++i) trsm(i, k);
    trsm(Y, k);
    for (int r = 0; r < k; ++r) trsm(wt(r), k);
    if (k + 1 < nb) {
      if (last) {
        column(k + 1, g0, gend - g0, true);
        if (k + 2 < nb) column(k + 2, g0, gend - g0, false);
        far_g0 = g0;
        far_kun = gend - g0;
        far_from = k + 3;
      } else {
        column(k + 1, g0, k + 1 - g0, true);
      }
    }
  }
}

}  // namespace elfihip
