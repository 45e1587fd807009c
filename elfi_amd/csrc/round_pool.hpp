// A pool of host threads for rounds of independent work items (pure C++: compiled with g++ by tests/test_lbfgsb.py).
#pragma once

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include <unistd.h>

namespace elfihip {

// Workers that live for ONE lcb_minimize call.  Each round the caller publishes a job (an index count and a function of
// the index) as a new GENERATION; every worker takes part in every generation exactly once -- it pulls indices from a
// shared counter until none are left and then checks out -- and the caller, which pulls indices as well, returns when
// all have checked out, so no worker is ever inside a job that is being replaced.  Between rounds (the device evaluates
// the points: a few hundred microseconds) workers poll the generation for up to a millisecond before they block on the
// condition variable: a wake-up through the kernel costs 50-100 us per round and thread, as long as the round's work
// (measured: 256 starts, 8 threads, 0.145 ms of host time per round against 0.06 ms of work).
class RoundPool {
 public:
  explicit RoundPool(int nthreads) : pid_(getpid()), nthreads_(nthreads) {
    for (int i = 1; i < nthreads; ++i) th_.emplace_back([this] { loop(); });
  }
  ~RoundPool() {
    if (getpid() != pid_) {   // a forked child inherited the thread OBJECTS, not the threads: nothing to stop or join
      for (auto& t : th_) t.detach();
      return;
    }
    {
      std::lock_guard<std::mutex> lock(mu_);
      stop_.store(true, std::memory_order_release);
    }
    cv_go_.notify_all();
    for (auto& t : th_) t.join();
  }
  int threads() const { return nthreads_; }
  // The pool was made by this process (a child forked after the pool's first use must not wait for workers it does not have).
  bool alive() const { return getpid() == pid_; }
  template <class F>
  void run(int64_t n, F f) {
    if (th_.empty() || n < 32 || !alive()) {
      for (int64_t i = 0; i < n; ++i) f(i);
      return;
    }
    // an exception inside an item (bad_alloc in a state machine) is carried to the caller instead of ending the process
    // from a worker thread; the remaining items of the round are skipped
    failed_.store(false, std::memory_order_relaxed);
    error_ = nullptr;
    fn_ = [this, &f](int64_t i) {
      if (failed_.load(std::memory_order_relaxed)) return;
      try {
        f(i);
      } catch (...) {
        std::lock_guard<std::mutex> lock(mu_);
        if (!error_) error_ = std::current_exception();
        failed_.store(true, std::memory_order_relaxed);
      }
    };
    n_ = n;
    next_.store(0, std::memory_order_relaxed);
    busy_.store((int)th_.size(), std::memory_order_relaxed);
    gen_.fetch_add(1, std::memory_order_release);
    {
      std::lock_guard<std::mutex> lock(mu_);   // (a worker about to block re-checks the generation under this lock)
      if (sleepers_ > 0) cv_go_.notify_all();
    }
    work();
    for (unsigned spin = 0; busy_.load(std::memory_order_acquire) != 0; ++spin)
      if ((spin & 63u) == 63u) std::this_thread::yield();
    if (error_) std::rethrow_exception(error_);
  }

 private:
  void work() {
    // one item per pull: items differ by an order of magnitude (a state machine that starts a new quasi-Newton
    // iteration against one that takes a line-search step), chunks of eight left threads idle at the end of a round
    for (;;) {
      const int64_t i = next_.fetch_add(1, std::memory_order_relaxed);
      if (i >= n_) return;
      fn_(i);
    }
  }
  void loop() {
    unsigned long long seen = 0;
    for (;;) {
      // poll for about a millisecond, then block
      const auto t0 = std::chrono::steady_clock::now();
      bool go = false;
      for (unsigned spin = 0;; ++spin) {
        if (gen_.load(std::memory_order_acquire) != seen) {
          go = true;
          break;
        }
        if (stop_.load(std::memory_order_acquire)) return;
        if ((spin & 255u) == 255u) {
          if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(1000)) break;
          std::this_thread::yield();
        }
      }
      if (!go) {
        std::unique_lock<std::mutex> lock(mu_);
        ++sleepers_;
        cv_go_.wait(lock, [&] { return stop_.load(std::memory_order_acquire) || gen_.load(std::memory_order_acquire) != seen; });
        --sleepers_;
        if (gen_.load(std::memory_order_acquire) == seen) return;   // stop
      }
      seen = gen_.load(std::memory_order_acquire);
      work();
      busy_.fetch_sub(1, std::memory_order_release);
    }
  }
  const pid_t pid_;
  const int nthreads_;
  std::vector<std::thread> th_;
  std::exception_ptr error_;
  std::atomic<bool> failed_{false};
  std::mutex mu_;
  std::condition_variable cv_go_;
  std::function<void(int64_t)> fn_;
  int64_t n_ = 0;
  std::atomic<int64_t> next_{0};
  std::atomic<unsigned long long> gen_{0};
  std::atomic<int> busy_{0};
  std::atomic<bool> stop_{false};
  int sleepers_ = 0;   // guarded by mu_
};

}  // namespace elfihip
