// host_threads.hpp -- the host side's few data-parallel loops (descriptor parsing over 10^4..10^5 pods) on a handful of threads.
// No pool: the threads live for one call (a spawn is ~20 us against loops of milliseconds), nothing outlives the entry point.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <exception>
#include <thread>
#include <vector>

namespace kg {

// threads a host loop may use: KG_HOST_THREADS, else min(8, hardware threads)
inline unsigned host_threads() {
  static const unsigned n = [] {
    const char* e = getenv("KG_HOST_THREADS");
    long v = e ? atol(e) : (long)std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
    return (unsigned)std::max(1l, std::min(v, 64l));
  }();
  return n;
}

// fewest items a chunk of a host loop is worth a thread for: KG_HOST_CHUNK_MIN (tests set it to a handful so that small inputs
// take the threaded path), else 2 048
inline unsigned host_chunk_min() {
  static const unsigned n = [] {
    const char* e = getenv("KG_HOST_CHUNK_MIN");
    return (unsigned)std::max(1l, e ? atol(e) : 2048l);
  }();
  return n;
}
// chunks for a loop over n items
inline unsigned host_chunks(size_t n) { return (unsigned)std::max<size_t>(1, std::min<size_t>(host_threads(), n / host_chunk_min())); }

// fn(c) for c in [0, chunks), chunk 0 on the calling thread.  An exception leaves through the caller: the one of the lowest
// chunk that threw (what a serial loop over the chunks would have met first).  A thread that cannot be started runs inline.
template <class F>
inline void parallel_chunks(unsigned chunks, F&& fn) {
  if (chunks <= 1) {
    if (chunks) fn(0u);
    return;
  }
  std::vector<std::exception_ptr> err(chunks);
  auto run = [&](unsigned c) {
    try {
      fn(c);
    } catch (...) {
      err[c] = std::current_exception();
    }
  };
  std::vector<std::thread> th;
  th.reserve(chunks - 1);
  std::vector<unsigned> inline_chunks;
  for (unsigned c = 1; c < chunks; ++c) {
    try {
      th.emplace_back(run, c);
    } catch (...) {
      inline_chunks.push_back(c);
    }
  }
  run(0u);
  for (unsigned c : inline_chunks) run(c);
  for (auto& t : th) t.join();
  for (auto& e : err)
    if (e) std::rethrow_exception(e);
}

}  // namespace kg
