// Host side of the fused factorisation sweep (gp_fit.hip: sweep_fused / step_kernel): WHICH tiles of the trailing
// matrix receive WHICH columns of the factor in WHICH step, and which workgroup does it.
//
// The right-looking sweep has a fixed chain per step -- panel solve, diagonal tile, diagonal block (about 50 us) --
// and a trailing update whose natural size falls from nb^2 / 2 tile updates in the first step to a handful in the
// last: updating "everything that can be updated" makes the first third of the steps update-bound (3 units per
// workgroup, 63 us, measured) and leaves the last third with idle compute units.  But a block column c is only READ
// when it becomes the panel (step c), so the columns of the factor it still has to receive can wait.  The schedule
// below gives every step the same update time T -- about what the diagonal block takes anyway, on the 255 compute
// units it leaves free -- and moves the excess of the early steps to the late ones, where it arrives as deeper updates
// (K up to 416 instead of 128-256: the tile's values move once).
//
// Granularity: the k range of the factor in 32-column "k-tiles" (what the kernel stages per barrier); a unit is the
// upper or lower 64 rows of a 128 x 128 tile receiving a contiguous range of k-tiles.  Block column c receives
// k-tiles [0, 4 (c-1)) from "far" units by the end of step c-2 and panel c-1 (k-tiles [4 (c-1), 4 c)) from "near"
// units in step c-1 (its diagonal tile from the launch before).  A tile half belongs to ONE workgroup per step, so the
// depth a column can absorb per step is bounded by T as well.
//
// Construction, BACKWARDS in time (as late as possible): going from the last step to the first, block column c holds
// rem[c] k-tiles not yet placed (the top of its range is placed first, i.e. into the latest steps).  At step k the
// k-tiles with index >= 4 k cannot be placed any earlier (panel k is solved in step k): they are mandatory.  After
// that every column's take is extended -- highest remaining index first, up to ktmax k-tiles -- as long as the step
// still packs into nwg workgroups of length T.  Forward greedy rules (earliest deadline first) leave the last block
// columns for the end, where their backlog can only be worked off one tile half per workgroup: 100-300 us steps; built
// backwards, what does not fit lands in the FIRST steps, where little of the factor exists and units are small.
// Rows of a column's tiles: Cholesky rows i >= c, the y block, and the L^-T rows r <= (last panel applied): an L^-T
// tile (r, c) is CREATED (C = -P P^T) by the unit that brings k-tile 4 r (block (r, r) is the first non-zero block of
// that row) and accumulates afterwards; its k range starts at 4 r.
// Host-only header (no HIP): tests/native/sweep_sched_check.cpp compiles it with g++ and replays every schedule.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <map>
#include <queue>
#include <utility>
#include <vector>

namespace elfihip {

struct SweepUnit {   // 16 bytes, read by step_kernel
  int32_t row;       // 0..nb-1: row block of A; nb: the y block; nb+1+r: row block r of WT (L^-T)
  int16_t c;         // block column of the tile
  int16_t kt0;       // first k-tile (column 32 kt0 of the factor)
  int16_t nkt;       // k-tiles (>= 2: the kernel keeps two in flight)
  uint8_t half;      // upper / lower 64 rows
  uint8_t keep;      // 1: C -= P P^T;  0: C = -P P^T
  int32_t pad;       // first unit of a workgroup: the number of its units in this step (else 0)
};
static_assert(sizeof(SweepUnit) == 16, "unit record layout");

struct SweepStep {
  int off0;          // index of this step's first workgroup offset in wg_off (nwg + 1 entries follow)
  int nwg;           // update workgroups with work
  int nunits;
  double makespan;   // predicted, microseconds
};

struct SweepSchedule {
  int nb = 0, nwg = 0, ktmax = 0;
  double target = 0.0, predicted_us = 0.0;
  std::vector<SweepUnit> units;    // all steps, workgroup by workgroup
  std::vector<int32_t> wg_off;     // per step nwg + 1 absolute offsets into `units`
  std::vector<SweepStep> steps;    // nb - 1 of them
};

// cost model (microseconds), measured on MI355X with scripts/native/step_probe.hip (248 workgroups, 1 / 2 / 4 units each,
// 2 ... 52 k-tiles deep): a workgroup needs 7 us before its first unit multiplies (launch, offset and unit records,
// the first operands), a unit 2.1 us (old and new C values, refilling the pipeline) plus 2.05 us per k-tile.  The
// chain of a step: the diagonal block beside the update 35 us, panel solve and diagonal tile 16.
constexpr double SWEEP_WG_START_US = 7.0;
constexpr double SWEEP_UNIT_FIXED_US = 2.1;
constexpr double SWEEP_UNIT_KT_US = 2.05;
constexpr double SWEEP_POTF2_US = 35.0;
constexpr double SWEEP_CHAIN_US = 16.0;

inline double sweep_unit_cost(int nkt) { return SWEEP_UNIT_FIXED_US + SWEEP_UNIT_KT_US * nkt; }

// the units of one touch: k-tiles [kt0, kt1) to block column c
inline void sweep_touch_units(int nb, int c, int kt0, int kt1, bool near, std::vector<SweepUnit>& out) {
  auto emit = [&](int row, int s, int n, int keep) {
    for (int h = 0; h < 2; ++h) {
      SweepUnit u;
      u.row = row;
      u.c = (int16_t)c;
      u.kt0 = (int16_t)s;
      u.nkt = (int16_t)n;
      u.half = (uint8_t)h;
      u.keep = (uint8_t)keep;
      u.pad = 0;
      out.push_back(u);
    }
  };
  for (int i = near ? c + 1 : c; i < nb; ++i) emit(i, kt0, kt1 - kt0, 1);
  emit(nb, kt0, kt1 - kt0, 1);
  const int jmax = (kt1 - 1) / 4;
  for (int r = 0; r <= jmax; ++r) {
    int s = std::max(kt0, 4 * r);
    const int keep = 4 * r >= kt0 ? 0 : 1;
    if (kt1 - s < 2) s = kt1 - 2;   // the kernel wants two k-tiles: one more from the zero blocks left of (r, r)
    emit(nb + 1 + r, s, kt1 - s, keep);
  }
}

constexpr int SWEEP_MAX_NKT = 512;

// Workgroups a multiset of units (count per nkt) needs when none may run longer than T: bins are filled largest
// unit first and identical bins are formed in bulk, so the cost depends on the number of distinct sizes only.
inline int sweep_bins_needed(const std::vector<int>& count, int max_nkt, double T) {
  static thread_local std::vector<int> left, pat;
  left.assign(count.begin(), count.begin() + max_nkt + 1);
  pat.assign(max_nkt + 1, 0);
  int bins = 0, top = max_nkt;
  for (;;) {
    while (top >= 2 && left[top] == 0) --top;
    if (top < 2) return bins;
    double rem = T;
    int times = 1 << 30;
    bool any = false;
    for (int n = top; n >= 2; --n) {
      pat[n] = 0;
      if (left[n] == 0) continue;
      const double cu = sweep_unit_cost(n);
      int m = (int)(rem / cu);
      if (!any && m == 0) m = 1;   // a unit longer than T still needs its workgroup
      if (m > left[n]) m = left[n];
      if (m == 0) continue;
      any = true;
      pat[n] = m;
      rem -= m * cu;
      times = std::min(times, left[n] / m);
    }
    for (int n = top; n >= 2; --n) left[n] -= times * pat[n];
    bins += times;
  }
}

// One construction of the sweep's schedule: steps of length T, no touch deeper than ktmax k-tiles (mandatory ones
// excepted).  Returns the predicted time; fills `S` when given.
inline double sweep_simulate(int nb, int nwg, double T, int ktmax, SweepSchedule* S, bool far_first = false) {
  std::vector<int> rem(nb + 2, 0), take(nb + 2, 0);
  for (int c = 2; c < nb; ++c) rem[c] = 4 * (c - 1);
  const int nsteps = std::max(0, nb - 1);
  std::vector<std::vector<SweepUnit>> step_units(S ? nsteps : 0);
  std::vector<double> makespan(nsteps, 0.0);
  std::vector<std::vector<int32_t>> step_off(S ? nsteps : 0);
  std::vector<int> step_active(nsteps, 0);
  std::vector<SweepUnit> units, tu, tu2;
  std::vector<double> loads(nwg);
  std::vector<int> count(SWEEP_MAX_NKT + 1), order;
  double total = 0.0;
  for (int k = nsteps - 1; k >= 0; --k) {
    units.clear();
    std::fill(count.begin(), count.end(), 0);
    int max_nkt = 4;
    auto tally = [&](const std::vector<SweepUnit>& us, int sign) {
      for (const SweepUnit& u : us) count[u.nkt] += sign;
    };
    tu.clear();
    sweep_touch_units(nb, k + 1, 4 * k, 4 * k + 4, true, tu);
    tally(tu, 1);
    units.insert(units.end(), tu.begin(), tu.end());
    // mandatory: the k-tiles that do not exist before step k
    for (int c = k + 2; c < nb; ++c) {
      take[c] = 0;
      if (rem[c] <= 0) continue;
      int x = std::max(0, rem[c] - 4 * k);
      if (x > 0) {
        if (rem[c] - x == 1) ++x;
        if (x == 1) x = 2;
        x = std::min(x, rem[c]);
        tu.clear();
        sweep_touch_units(nb, c, rem[c] - x, rem[c], false, tu);
        tally(tu, 1);
        take[c] = x;
        max_nkt = std::max(max_nkt, x);
      }
    }
    // extensions, highest remaining index first (the next to become mandatory)
    order.clear();
    for (int c = k + 2; c < nb; ++c)
      if (rem[c] > 0) order.push_back(c);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return rem[a] > rem[b]; });
    for (int c : order) {
      const int x0 = take[c];
      int x = std::min(rem[c], std::max(ktmax, x0));
      if (rem[c] - x == 1) --x;
      tu.clear();
      if (x0) sweep_touch_units(nb, c, rem[c] - x0, rem[c], false, tu);
      while (x > x0 && x >= 2) {
        tu2.clear();
        sweep_touch_units(nb, c, rem[c] - x, rem[c], false, tu2);
        tally(tu, -1);
        tally(tu2, 1);
        if (sweep_bins_needed(count, std::max(max_nkt, x), T - SWEEP_WG_START_US) <= nwg) {
          take[c] = x;
          max_nkt = std::max(max_nkt, x);
          break;
        }
        tally(tu2, -1);
        tally(tu, 1);
        x = std::max(x0, x - x0 <= 6 ? x - 2 : x0 + (x - x0) / 2);
        if (rem[c] - x == 1) --x;
      }
    }
    for (int c = k + 2; c < nb; ++c)
      if (take[c] > 0) {
        sweep_touch_units(nb, c, rem[c] - take[c], rem[c], false, units);
        rem[c] -= take[c];
      }
    // the deal, bin by bin exactly as sweep_bins_needed counts: a workgroup is filled largest unit first with whatever
    // still fits into T; what is left when the workgroups run out goes to the emptiest ones
    std::vector<std::vector<int>> mine(nwg);
    std::fill(loads.begin(), loads.end(), 0.0);
    double worst = 0.0;
    {
      std::vector<int> ord(units.size());
      for (size_t i = 0; i < units.size(); ++i) ord[i] = (int)i;
      std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return units[a].nkt > units[b].nkt; });
      std::vector<char> placed(units.size(), 0);
      size_t first = 0;
      for (int w = 0; w < nwg && first < ord.size(); ++w) {
        double room = T - SWEEP_WG_START_US + 1e-9;
        bool any = false;
        for (size_t p = first; p < ord.size(); ++p) {
          const int i = ord[p];
          if (placed[i]) continue;
          const double cu = sweep_unit_cost(units[i].nkt);
          if (cu <= room || !any) {
            mine[w].push_back(i);
            loads[w] += cu;
            placed[i] = 1;
            room -= cu;
            any = true;
            if (room < sweep_unit_cost(2)) break;
          }
        }
        while (first < ord.size() && placed[ord[first]]) ++first;
      }
      for (size_t p = first; p < ord.size(); ++p) {
        const int i = ord[p];
        if (placed[i]) continue;
        int w = 0;
        for (int x = 1; x < nwg; ++x)
          if (loads[x] < loads[w]) w = x;
        mine[w].push_back(i);
        loads[w] += sweep_unit_cost(units[i].nkt);
      }
      for (int w = 0; w < nwg; ++w) worst = std::max(worst, loads[w]);
      worst += SWEEP_WG_START_US;
    }
    makespan[k] = worst;
    total += std::max(SWEEP_POTF2_US, worst) + SWEEP_CHAIN_US;
    if (S) {
      // the longest workgroups first, empty ones last; a workgroup's units by column, row, half: neighbours share
      // operands.  far_first (the chained form of the step launch): the LAST columns first -- the units of column k+1
      // and k+2 read the panel solved inside the same launch and may have to wait for it
      std::vector<int> wgs(nwg);
      for (int w = 0; w < nwg; ++w) wgs[w] = w;
      std::stable_sort(wgs.begin(), wgs.end(), [&](int a, int b) { return loads[a] > loads[b]; });
      std::vector<SweepUnit>& out = step_units[k];
      std::vector<int32_t>& off = step_off[k];
      for (int w : wgs) {
        off.push_back((int32_t)out.size());
        std::vector<int>& m = mine[w];
        std::sort(m.begin(), m.end(), [&](int a, int b) {
          const SweepUnit &x = units[a], &y = units[b];
          if (x.c != y.c) return far_first ? x.c > y.c : x.c < y.c;
          if (x.row != y.row) return x.row < y.row;
          return x.half < y.half;
        });
        for (int i : m) out.push_back(units[i]);
        if (!m.empty()) {
          ++step_active[k];
          out[out.size() - m.size()].pad = (int32_t)m.size();   // a workgroup's first unit carries its unit count
        }
      }
      off.push_back((int32_t)out.size());
    }
  }
  if (S) {
    S->nb = nb;
    S->nwg = nwg;
    S->target = T;
    S->ktmax = ktmax;
    S->predicted_us = total;
    S->units.clear();
    S->wg_off.clear();
    S->steps.clear();
    for (int k = 0; k < nsteps; ++k) {
      SweepStep st;
      st.off0 = (int)S->wg_off.size();
      st.nwg = step_active[k];
      st.nunits = (int)step_units[k].size();
      st.makespan = makespan[k];
      const int32_t base = (int32_t)S->units.size();
      for (int32_t o : step_off[k]) S->wg_off.push_back(base + o);
      S->units.insert(S->units.end(), step_units[k].begin(), step_units[k].end());
      S->steps.push_back(st);
    }
  }
  return total;
}

// The schedule for nb block columns on nwg update workgroups: the best T of a few by predicted time, from the longer of
// the diagonal block's time and the average update time per step upwards; ktmax = the deepest unit that fits into T.
inline void sweep_build(int nb, int nwg, SweepSchedule* S, bool far_first = false) {
  double work = 0.0;   // k-tiles of half-tile units
  for (int c = 1; c < nb; ++c) {
    work += 2.0 * ((double)(nb - c + 1) * 4 * c - 4);
    for (int r = 0; r < c; ++r) work += 2.0 * (4 * c - 4 * r);
  }
  const double avg = SWEEP_WG_START_US + 1.12 * work * SWEEP_UNIT_KT_US / ((double)std::max(1, nb - 1) * nwg);
  const double T0 = std::max(SWEEP_POTF2_US, avg);
  // where the update sets the step length by a wide margin (many block columns) the packing is loose enough for one try
  static const double F[] = {1.0, 1.03, 1.06, 1.12};
  const int ntry = avg > 2.0 * SWEEP_POTF2_US ? 1 : 4;
  double best = -1.0, bT = T0;
  int bk = std::min(SWEEP_MAX_NKT, std::max(4, (int)((T0 - SWEEP_WG_START_US - SWEEP_UNIT_FIXED_US) / SWEEP_UNIT_KT_US)));
  for (int fi = 0; fi < ntry && ntry > 1; ++fi) {
    const double f = F[fi];
    const double T = T0 * f;
    const int big = std::min(SWEEP_MAX_NKT, std::max(4, (int)((T - SWEEP_WG_START_US - SWEEP_UNIT_FIXED_US) / SWEEP_UNIT_KT_US)));
    const double t = sweep_simulate(nb, nwg, T, big, nullptr);
    if (best < 0.0 || t < best) {
      best = t;
      bT = T;
      bk = big;
    }
  }
  sweep_simulate(nb, nwg, bT, bk, S, far_first);
}
// Replay a schedule symbolically; returns 0 when every tile half receives exactly the k range it must, in order, in
// time, created once, and no two workgroups touch the same tile half in one step.  (tests)
inline int sweep_check(const SweepSchedule& S, char* msg, size_t msglen) {
  const int nb = S.nb;
  std::map<long long, int> have;   // (row, c, half) -> k-tiles applied so far (exclusive end)
  auto fail = [&](const char* what, int k, const SweepUnit& u) {
    std::snprintf(msg, msglen, "%s: step %d unit row %d c %d kt0 %d nkt %d half %d keep %d", what, k, u.row, u.c, u.kt0,
                  u.nkt, u.half, u.keep);
    return 1;
  };
  if ((int)S.steps.size() != nb - 1) {
    std::snprintf(msg, msglen, "steps %d for nb %d", (int)S.steps.size(), nb);
    return 1;
  }
  for (int k = 0; k + 1 < nb; ++k) {
    const SweepStep& st = S.steps[k];
    std::map<long long, int> wg_of;
    for (int w = 0; w < S.nwg; ++w) {
      const int lo = S.wg_off[st.off0 + w], hi = S.wg_off[st.off0 + w + 1];
      if ((w >= st.nwg) != (lo == hi)) {
        std::snprintf(msg, msglen, "step %d: workgroup %d of %d active is %s", k, w, st.nwg, lo == hi ? "empty" : "not empty");
        return 1;
      }
      for (int i = lo; i < hi; ++i) {
        const SweepUnit& u = S.units[i];
        const long long key = (((long long)u.row * 4096 + u.c) << 1) | u.half;
        if (u.nkt < 2 || u.kt0 < 0) return fail("k range too short", k, u);
        if (u.kt0 + u.nkt > 4 * (k + 1)) return fail("panel not solved yet", k, u);
        if (u.c <= k || u.c >= nb) return fail("block column already final", k, u);
        if (u.row < nb && u.row < u.c) return fail("upper triangle", k, u);
        if (u.row == u.c && u.kt0 + u.nkt > 4 * (u.c - 1)) return fail("diagonal tile belongs to the launch before", k, u);
        auto w_it = wg_of.find(key);
        if (w_it != wg_of.end() && w_it->second != w) return fail("tile half on two workgroups", k, u);
        wg_of[key] = w;
        const int start = u.row > nb ? 4 * (u.row - nb - 1) : 0;
        auto it = have.find(key);
        if (it == have.end()) {
          if (u.row > nb) {
            if (u.keep != 0) return fail("L^-T tile accumulated before it was created", k, u);
            if (u.kt0 > start || u.kt0 + u.nkt <= start) return fail("L^-T tile created without its first panel", k, u);
          } else {
            if (u.keep != 1 || u.kt0 != 0) return fail("first update does not start at 0", k, u);
          }
        } else {
          if (u.keep != 1) return fail("tile overwritten", k, u);
          if (u.kt0 != it->second) return fail("k range not contiguous", k, u);
        }
        have[key] = u.kt0 + u.nkt;
      }
    }
    // after step k block column k+1 is complete (all but its diagonal tile) and so is column k+2 up to panel k
    for (int c = k + 1; c <= std::min(k + 2, nb - 1); ++c) {
      const int need = c == k + 1 ? 4 * c : 4 * (c - 1);
      for (int row = (c == k + 1 ? c + 1 : c); row <= nb + 1 + std::min(k, nb); ++row) {
        if (row > nb && 4 * (row - nb - 1) >= need) continue;
        for (int h = 0; h < 2; ++h) {
          const long long key = (((long long)row * 4096 + c) << 1) | h;
          auto it = have.find(key);
          if (it == have.end() || it->second != need) {
            std::snprintf(msg, msglen, "after step %d: tile (%d, %d) half %d has %d of %d k-tiles", k, row, c, h,
                          it == have.end() ? -1 : it->second, need);
            return 1;
          }
        }
      }
    }
  }
  return 0;
}

}  // namespace elfihip
