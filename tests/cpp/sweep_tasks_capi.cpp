// Test-only C wrapper around elfi_amd/csrc/sweep_tasks.hpp (pure C++), compiled with g++ by tests/test_sweep_tasks.py.
#include <cstddef>

#include "../../elfi_amd/csrc/sweep_tasks.hpp"

extern "C" {

// number of tasks for nb block columns in panel groups of G; if `out` is not NULL it receives 10 ints per task
int sweep_tasks(int nb, int G, int* out) {
  std::vector<elfihip::SweepTask> v;
  elfihip::sweep_build_tasks(nb, G, &v);
  if (out)
    for (std::size_t i = 0; i < v.size(); ++i) {
      const elfihip::SweepTask& t = v[i];
      const int f[10] = {t.type, t.rb, t.c, t.k0, t.kun, t.prior, t.need_pdone, t.need_rb, t.need_c, t.beta0};
      for (int j = 0; j < 10; ++j) out[10 * i + j] = f[j];
    }
  return (int)v.size();
}
}
