// Test-only C wrapper around elfi_amd/csrc/sweep_tasks.hpp (pure C++), compiled with g++ by tests/test_sweep_tasks.py.
#include <cstddef>

#include "../../elfi_amd/csrc/sweep_tasks.hpp"

extern "C" {

// number of tasks for nb block columns; if `out` is not NULL it receives 4 ints per task (type, rb, c, k)
int sweep_tasks(int nb, int* out) {
  std::vector<elfihip::SweepTask> v;
  elfihip::sweep_build_tasks(nb, &v);
  if (out)
    for (std::size_t i = 0; i < v.size(); ++i) {
      out[4 * i] = v[i].type;
      out[4 * i + 1] = v[i].rb;
      out[4 * i + 2] = v[i].c;
      out[4 * i + 3] = v[i].k;
    }
  return (int)v.size();
}
}
