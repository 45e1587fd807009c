// Replays the fused sweep's update schedule for every block-column count the fused schedule serves (host-only check of
// elfi_amd/csrc/sweep_sched.hpp: every tile half receives exactly its k range, in order, in time).
//   g++ -O2 -std=c++17 -I elfi_amd/csrc -o /tmp/sweep_sched_check tests/native/sweep_sched_check.cpp && /tmp/sweep_sched_check
#include "sweep_sched.hpp"

#include <cstdio>
#include <cstdlib>

int main(int argc, char** argv) {
  using namespace elfihip;
  const int lo = argc > 1 ? std::atoi(argv[1]) : 2, hi = argc > 2 ? std::atoi(argv[2]) : 64;
  const int nwg = argc > 3 ? std::atoi(argv[3]) : 248;
  int bad = 0;
  for (int nb = lo; nb <= hi; ++nb) {
    SweepSchedule S;
    sweep_build(nb, nwg, &S);
    char msg[256] = "";
    const int rc = sweep_check(S, msg, sizeof msg);
    double worst = 0.0;
    for (const SweepStep& st : S.steps) worst = st.makespan > worst ? st.makespan : worst;
    std::printf("nb %2d  T %.1f ktmax %2d  units %6zu  predicted %7.0f us  longest step %5.1f  %s %s\n", nb, S.target, S.ktmax,
                S.units.size(), S.predicted_us, worst, rc ? "FAIL" : "ok", msg);
    if (argc > 4) {
      for (const SweepStep& st : S.steps) std::printf(" %.0f/%d", st.makespan, st.nunits);
      std::printf("\n");
    }
    bad += rc;
  }
  return bad ? 1 : 0;
}
