// host_threads.h -- the few places where the host side of the constructor works side by side (the parts of a large g2o
// file, the halves of a dissection, the candidate analyses): never more threads than the host has cores, every thread
// joined whatever happens, an exception of a worker rethrown on the calling thread, and a thread that cannot be created
// (EAGAIN under a container's limit) simply not used -- its work runs on the caller.  Nothing here outlives the call.
#pragma once
#include <algorithm>
#include <atomic>
#include <exception>
#include <system_error>
#include <thread>
#include <vector>

namespace rrpgo {

// fn(i) for i in [0, n), on at most min(n, max_threads, hardware cores) threads (the caller's included); indices are
// drawn from one counter.  The first exception (lowest index) is rethrown after every thread has been joined.
template <class F> void parallel_indices(int n, int max_threads, F &&fn) {
  if (n <= 0) return;
  const int hw = (int)std::max(1u, std::thread::hardware_concurrency());
  const int workers = std::max(1, std::min(std::min(n, max_threads), hw));
  std::atomic<int> next{0};
  std::vector<std::exception_ptr> err((size_t)n);
  auto body = [&] {
    for (int i; (i = next.fetch_add(1)) < n;) {
      try { fn(i); } catch (...) { err[(size_t)i] = std::current_exception(); }
    }
  };
  std::vector<std::thread> pool;
  struct Join { std::vector<std::thread> &p; ~Join() { for (std::thread &t : p) if (t.joinable()) t.join(); } } join{pool};
  try {
    pool.reserve((size_t)workers);
    for (int k = 1; k < workers; k++) pool.emplace_back(body);
  } catch (const std::system_error &) {
    // no more threads to be had: the ones that started and the caller share the indices
  }
  body();
  for (std::thread &t : pool) t.join();
  for (const std::exception_ptr &e : err) if (e) std::rethrow_exception(e);
}

// `here` on the calling thread, `other` beside it (after it when no thread can be had); both complete before an exception
// of either is rethrown
template <class F, class G> void run_beside(F &&other, G &&here) {
  parallel_indices(2, 2, [&](int i) { if (i == 0) here(); else other(); });
}

}  // namespace rrpgo
