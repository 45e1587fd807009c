// A pool of host threads for rounds of independent work items (pure C++: compiled with g++ by tests/test_lbfgsb.py).
#pragma once

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace elfihip {

// Workers that live for ONE lcb_minimize call.  Each round the caller publishes a job (an index count and a function of
// the index) as a new GENERATION; every worker takes part in every generation exactly once -- it pulls index chunks from
// a shared counter until none are left and then checks out -- and the caller, which pulls chunks as well, returns when
// all have checked out.  (A wake-up through the condition variable costs some 10 us per round against rounds of 1 ms.)
class RoundPool {
 public:
  explicit RoundPool(int nthreads) {
    for (int i = 1; i < nthreads; ++i) th_.emplace_back([this] { loop(); });
  }
  ~RoundPool() {
    {
      std::lock_guard<std::mutex> lock(mu_);
      stop_ = true;
    }
    cv_go_.notify_all();
    for (auto& t : th_) t.join();
  }
  template <class F>
  void run(int64_t n, F f) {
    if (th_.empty() || n < 32) {
      for (int64_t i = 0; i < n; ++i) f(i);
      return;
    }
    {
      std::lock_guard<std::mutex> lock(mu_);
      fn_ = [&f](int64_t i) { f(i); };
      n_ = n;
      next_.store(0, std::memory_order_relaxed);
      busy_ = (int)th_.size();
      ++gen_;
    }
    cv_go_.notify_all();
    work();
    std::unique_lock<std::mutex> lock(mu_);
    cv_done_.wait(lock, [this] { return busy_ == 0; });
  }

 private:
  void work() {
    for (;;) {
      const int64_t i0 = next_.fetch_add(8, std::memory_order_relaxed);
      if (i0 >= n_) return;
      const int64_t i1 = std::min<int64_t>(i0 + 8, n_);
      for (int64_t i = i0; i < i1; ++i) fn_(i);
    }
  }
  void loop() {
    unsigned long long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lock(mu_);
        cv_go_.wait(lock, [&] { return stop_ || gen_ != seen; });
        if (stop_) return;
        seen = gen_;
      }
      work();
      {
        std::lock_guard<std::mutex> lock(mu_);
        if (--busy_ == 0) cv_done_.notify_one();
      }
    }
  }
  std::vector<std::thread> th_;
  std::mutex mu_;
  std::condition_variable cv_go_, cv_done_;
  std::function<void(int64_t)> fn_;
  int64_t n_ = 0;
  std::atomic<int64_t> next_{0};
  unsigned long long gen_ = 0;
  int busy_ = 0;
  bool stop_ = false;
};

}  // namespace elfihip
