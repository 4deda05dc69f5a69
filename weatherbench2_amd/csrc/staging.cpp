// Host -> HBM staging for chunks that arrive in PAGEABLE memory.
//
// The reference's Beam pipeline hands _evaluate_chunk NumPy-backed datasets
// (xbeam.DatasetToChunks, /root/reference/weatherbench2/evaluation.py:693-705):
// ordinary malloc'ed pages.  A DMA engine cannot read those, so every byte is
// copied once into page-locked memory on its way to the device.  One thread's
// memcpy runs at ~10 GB/s -- a sixth of what the PCIe 5.0 x16 link moves -- so
// the staging copy is done by a small pool of threads, slice by slice through a
// ring of pinned slots: while the DMA of slice k runs, the pool fills slice
// k + 1.  hipHostRegister of the source was the alternative; pinning fresh
// pages costs more than copying them (get_user_pages per 4 KiB page) and the
// arrays are new every chunk.
//
//   wb2_uploader_create / _destroy     ring + pool, once per feeder thread
//   wb2_uploader_upload                one pageable buffer -> device memory,
//                                      asynchronous on the caller's stream;
//                                      returns when the SOURCE may be reused
#include "common.hpp"
#include "trace.hpp"
#include "wb2hip.h"

#include <immintrin.h>

#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace wb2 {
namespace {

// One piece of a staging copy with NON-TEMPORAL stores: the destination (a
// pinned slot the DMA engine reads next) never needs to be in a cache, and a
// plain store would first fetch every destination line (read-for-ownership):
// 3 bytes of DRAM traffic per byte copied instead of 2.  glibc's memcpy only
// switches to streaming stores above ~3/4 of the shared cache size -- far more
// than one thread's piece of a slice.
__attribute__((target("avx2"))) void stream_copy_avx2(char* dst,
                                                      const char* src,
                                                      size_t n) {
  for (; n >= 128; n -= 128, src += 128, dst += 128) {
    const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src));
    const __m256i b =
        _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + 32));
    const __m256i c =
        _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + 64));
    const __m256i d =
        _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + 96));
    _mm256_stream_si256(reinterpret_cast<__m256i*>(dst), a);
    _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + 32), b);
    _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + 64), c);
    _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + 96), d);
  }
  _mm_sfence();
  if (n) std::memcpy(dst, src, n);
}

void stream_copy_sse2(char* dst, const char* src, size_t n) {
  for (; n >= 64; n -= 64, src += 64, dst += 64) {
    const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src));
    const __m128i b =
        _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + 16));
    const __m128i c =
        _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + 32));
    const __m128i d =
        _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + 48));
    _mm_stream_si128(reinterpret_cast<__m128i*>(dst), a);
    _mm_stream_si128(reinterpret_cast<__m128i*>(dst + 16), b);
    _mm_stream_si128(reinterpret_cast<__m128i*>(dst + 32), c);
    _mm_stream_si128(reinterpret_cast<__m128i*>(dst + 48), d);
  }
  _mm_sfence();
  if (n) std::memcpy(dst, src, n);
}

void stream_copy(char* dst, const char* src, size_t n) {
  static const bool plain = [] {  // WB2HIP_STAGE_MEMCPY=1: A/B runs
    const char* e = getenv("WB2HIP_STAGE_MEMCPY");
    return e && e[0] == '1';
  }();
  if (plain || n < 4096) {
    std::memcpy(dst, src, n);
    return;
  }
  // streaming stores want an aligned destination
  const size_t head = (size_t)(-reinterpret_cast<uintptr_t>(dst)) & 31;
  if (head) {
    std::memcpy(dst, src, head);
    dst += head, src += head, n -= head;
  }
  static const bool avx2 = __builtin_cpu_supports("avx2");
  if (avx2)
    stream_copy_avx2(dst, src, n);
  else
    stream_copy_sse2(dst, src, n);
}

// A fork-join pool: copy() cuts a buffer into one piece per thread (the caller
// takes piece 0) and returns when all pieces are done.
class CopyPool {
 public:
  explicit CopyPool(int n_threads) : n_(n_threads < 1 ? 1 : n_threads) {
    for (int i = 1; i < n_; ++i) workers_.emplace_back([this, i] { loop(i); });
  }
  ~CopyPool() {
    {
      std::lock_guard<std::mutex> lock(mu_);
      stop_ = true;
      ++generation_;
    }
    cv_.notify_all();
    for (auto& w : workers_) w.join();
  }
  int size() const { return n_; }

  void copy(char* dst, const char* src, size_t nbytes) {
    if (n_ == 1 || nbytes < (size_t)(1 << 20)) {
      stream_copy(dst, src, nbytes);
      return;
    }
    {
      std::lock_guard<std::mutex> lock(mu_);
      dst_ = dst;
      src_ = src;
      nbytes_ = nbytes;
      pending_ = n_ - 1;
      ++generation_;
    }
    cv_.notify_all();
    part(0);
    std::unique_lock<std::mutex> lock(mu_);
    done_.wait(lock, [this] { return pending_ == 0; });
  }

 private:
  void part(int i) {
    // 4 KiB-aligned cuts: no two threads share a destination page
    const size_t per = ((nbytes_ / n_) + 4095) & ~(size_t)4095;
    const size_t lo = per * i < nbytes_ ? per * i : nbytes_;
    const size_t hi = (i == n_ - 1) ? nbytes_
                                    : (lo + per < nbytes_ ? lo + per : nbytes_);
    if (hi > lo) stream_copy(dst_ + lo, src_ + lo, hi - lo);
  }
  void loop(int i) {
    unsigned long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lock(mu_);
        cv_.wait(lock, [&] { return generation_ != seen; });
        seen = generation_;
        if (stop_) return;
      }
      part(i);
      std::lock_guard<std::mutex> lock(mu_);
      if (--pending_ == 0) done_.notify_one();
    }
  }
  const int n_;
  std::vector<std::thread> workers_;
  std::mutex mu_;
  std::condition_variable cv_, done_;
  unsigned long generation_ = 0;
  bool stop_ = false;
  int pending_ = 0;
  char* dst_ = nullptr;
  const char* src_ = nullptr;
  size_t nbytes_ = 0;
};

struct Uploader {
  CopyPool pool;
  size_t slot_bytes;
  std::vector<char*> slots;
  std::vector<hipEvent_t> events;
  std::vector<bool> busy;
  int next = 0;
  // WB2HIP_DMA_STREAMS=n > 1: the DMAs of consecutive slices alternate between
  // n streams of the uploader's own.  Measured and not kept as the default
  // (profiles/r06_upload_sweep.txt): 53.6 GB/s on the caller's stream, 48.6
  // with two, 53.1 with three -- one stream keeps the link busy.
  std::vector<hipStream_t> dma;
  hipEvent_t fork = nullptr;
  std::vector<hipEvent_t> join;
  size_t slices = 0;
  Uploader(int n_threads, size_t slot) : pool(n_threads), slot_bytes(slot) {}
};

int dma_streams() {
  static const int n = [] {
    const char* e = getenv("WB2HIP_DMA_STREAMS");
    const int v = e ? atoi(e) : 1;
    return v < 1 ? 1 : (v > 4 ? 4 : v);
  }();
  return n;
}

}  // namespace
}  // namespace wb2

extern "C" {

int wb2_uploader_create(int32_t n_threads, int64_t slot_bytes, int32_t n_slots,
                        void** uploader_out) {
  WB2_TRACE();
  using namespace wb2;
  WB2_REQUIRE(uploader_out != nullptr, "null pointer argument");
  WB2_REQUIRE(n_threads >= 1 && n_threads <= 256, "n_threads=%d", n_threads);
  WB2_REQUIRE(slot_bytes >= 4096 && n_slots >= 2 && n_slots <= 64,
              "slot_bytes=%lld n_slots=%d", (long long)slot_bytes, n_slots);
  auto* up = new Uploader(n_threads, (size_t)slot_bytes);
  for (int i = 0; i < n_slots; ++i) {
    void* p = nullptr;
    hipEvent_t ev = nullptr;
    if (hipHostMalloc(&p, (size_t)slot_bytes, hipHostMallocDefault) !=
            hipSuccess ||
        hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
      if (p) (void)hipHostFree(p);
      for (char* s : up->slots) (void)hipHostFree(s);
      for (hipEvent_t e : up->events) (void)hipEventDestroy(e);
      delete up;
      return fail("pinned ring of %d x %lld bytes could not be allocated",
                  n_slots, (long long)slot_bytes);
    }
    // first touch by this thread: the pages exist before the first upload
    std::memset(p, 0, (size_t)slot_bytes);
    up->slots.push_back(static_cast<char*>(p));
    up->events.push_back(ev);
    up->busy.push_back(false);
  }
  if (dma_streams() > 1) {
    bool ok = hipEventCreateWithFlags(&up->fork, hipEventDisableTiming) ==
              hipSuccess;
    for (int i = 0; ok && i < dma_streams(); ++i) {
      hipStream_t st = nullptr;
      hipEvent_t ev = nullptr;
      ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess &&
           hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess;
      if (st) up->dma.push_back(st);
      if (ev) up->join.push_back(ev);
    }
    if (!ok) {
      wb2_uploader_destroy(up);
      return fail("the uploader's DMA streams could not be created");
    }
  }
  *uploader_out = up;
  return 0;
}

int wb2_uploader_destroy(void* uploader) {
  WB2_TRACE();
  using namespace wb2;
  if (!uploader) return 0;
  auto* up = static_cast<Uploader*>(uploader);
  for (size_t i = 0; i < up->slots.size(); ++i) {
    if (up->busy[i]) (void)hipEventSynchronize(up->events[i]);
    (void)hipEventDestroy(up->events[i]);
    (void)hipHostFree(up->slots[i]);
  }
  for (hipStream_t st : up->dma) {
    (void)hipStreamSynchronize(st);
    (void)hipStreamDestroy(st);
  }
  for (hipEvent_t ev : up->join) (void)hipEventDestroy(ev);
  if (up->fork) (void)hipEventDestroy(up->fork);
  delete up;
  return 0;
}

int wb2_host_copy(void* dst, const void* src, int64_t nbytes,
                  int32_t n_threads) {
  WB2_TRACE();
  using namespace wb2;
  WB2_EMPTY_OK(nbytes);
  WB2_REQUIRE(dst && src, "null pointer argument");
  WB2_REQUIRE(n_threads >= 1 && n_threads <= 256, "n_threads=%d", n_threads);
  CopyPool pool(n_threads);
  pool.copy(static_cast<char*>(dst), static_cast<const char*>(src),
            (size_t)nbytes);
  return 0;
}

namespace wb2 {
namespace {
int uploader_fork(Uploader* up, hipStream_t s);
int uploader_join(Uploader* up, hipStream_t s);
int upload_one(Uploader* up, void* dst, const void* src, int64_t nbytes,
               hipStream_t s);
}  // namespace
}  // namespace wb2

int wb2_uploader_upload(void* uploader, void* dst, const void* src,
                        int64_t nbytes, void* stream) {
  WB2_TRACE();
  using namespace wb2;
  WB2_REQUIRE(uploader != nullptr, "null uploader");
  WB2_EMPTY_OK(nbytes);
  WB2_REQUIRE(dst && src, "null pointer argument");
  auto* up = static_cast<Uploader*>(uploader);
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc = uploader_fork(up, s);
  if (rc == 0) rc = upload_one(up, dst, src, nbytes, s);
  const int rj = uploader_join(up, s);
  return rc != 0 ? rc : rj;
}

int wb2_uploader_upload_many(void* uploader, int32_t n, void* const* dst,
                             const void* const* src, const int64_t* nbytes,
                             void* stream) {
  WB2_TRACE();
  using namespace wb2;
  WB2_REQUIRE(uploader != nullptr, "null uploader");
  WB2_EMPTY_OK(n);
  WB2_REQUIRE(dst && src && nbytes, "null pointer argument");
  auto* up = static_cast<Uploader*>(uploader);
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc = uploader_fork(up, s);
  for (int i = 0; rc == 0 && i < n; ++i) {
    WB2_REQUIRE(nbytes[i] >= 0, "nbytes[%d]=%lld is negative", i,
                (long long)nbytes[i]);
    if (nbytes[i] == 0) continue;
    WB2_REQUIRE(dst[i] && src[i], "buffer %d is null", i);
    rc = upload_one(up, dst[i], src[i], nbytes[i], s);
  }
  const int rj = uploader_join(up, s);
  return rc != 0 ? rc : rj;
}

int wb2_uploader_download(void* uploader, void* dst, const void* src,
                          int64_t nbytes, void* stream) {
  WB2_TRACE();
  using namespace wb2;
  WB2_REQUIRE(uploader != nullptr, "null uploader");
  WB2_EMPTY_OK(nbytes);
  WB2_REQUIRE(dst && src, "null pointer argument");
  auto* up = static_cast<Uploader*>(uploader);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int n_slots = (int)up->slots.size();
  // slots still being read by the DMA of an earlier upload
  for (int i = 0; i < n_slots; ++i) {
    if (up->busy[i]) {
      WB2_HIP_OK(hipEventSynchronize(up->events[i]));
      up->busy[i] = false;
    }
  }
  const char* from = static_cast<const char*>(src);
  char* to = static_cast<char*>(dst);
  size_t left = (size_t)nbytes;
  // slices in flight, oldest first: slot (head + k) % n_slots holds slice k
  int head = up->next, in_flight = 0;
  char* out = to;
  std::vector<size_t> len((size_t)n_slots, 0);
  auto drain_one = [&]() -> int {
    WB2_HIP_OK(hipEventSynchronize(up->events[head]));
    up->pool.copy(out, up->slots[head], len[(size_t)head]);
    out += len[(size_t)head];
    head = (head + 1) % n_slots;
    --in_flight;
    return 0;
  };
  int rc = 0;
  while (rc == 0 && (left || in_flight)) {
    if (left && in_flight < n_slots) {
      const size_t n = left < up->slot_bytes ? left : up->slot_bytes;
      const int i = (head + in_flight) % n_slots;
      const hipError_t e1 =
          hipMemcpyAsync(up->slots[i], from, n, hipMemcpyDeviceToHost, s);
      const hipError_t e2 =
          e1 == hipSuccess ? hipEventRecord(up->events[i], s) : e1;
      if (e2 != hipSuccess) {
        // what is in flight writes into the ring: wait before the slots are
        // used again
        (void)hipStreamSynchronize(s);
        return fail("HIP error: %s", hipGetErrorString(e2));
      }
      len[(size_t)i] = n;
      ++in_flight;
      from += n;
      left -= n;
    } else {
      rc = drain_one();
    }
  }
  if (rc != 0) (void)hipStreamSynchronize(s);
  up->next = head;
  return rc;
}

}  // extern "C"

namespace wb2 {
namespace {
// The uploader's DMA streams start behind what the caller's stream holds (the
// destination may be a block that stream has just finished with) ...
int uploader_fork(Uploader* up, hipStream_t s) {
  if (up->dma.empty()) return 0;
  WB2_HIP_OK(hipEventRecord(up->fork, s));
  for (hipStream_t st : up->dma) WB2_HIP_OK(hipStreamWaitEvent(st, up->fork, 0));
  return 0;
}
// ... and the caller's stream goes on behind their copies.
int uploader_join(Uploader* up, hipStream_t s) {
  for (size_t i = 0; i < up->dma.size(); ++i) {
    WB2_HIP_OK(hipEventRecord(up->join[i], up->dma[i]));
    WB2_HIP_OK(hipStreamWaitEvent(s, up->join[i], 0));
  }
  return 0;
}

int upload_one(Uploader* up, void* dst, const void* src, int64_t nbytes,
               hipStream_t s) {
  const char* from = static_cast<const char*>(src);
  char* to = static_cast<char*>(dst);
  size_t left = (size_t)nbytes;
  while (left) {
    const size_t n = left < up->slot_bytes ? left : up->slot_bytes;
    const int i = up->next;
    up->next = (i + 1) % (int)up->slots.size();
    if (up->busy[i]) {  // the DMA that last read this slot
      WB2_HIP_OK(hipEventSynchronize(up->events[i]));
      up->busy[i] = false;
    }
    up->pool.copy(up->slots[i], from, n);
    hipStream_t ds = up->dma.empty() ? s : up->dma[up->slices++ % up->dma.size()];
    WB2_HIP_OK(hipMemcpyAsync(to, up->slots[i], n, hipMemcpyHostToDevice, ds));
    WB2_HIP_OK(hipEventRecord(up->events[i], ds));
    up->busy[i] = true;
    from += n;
    to += n;
    left -= n;
  }
  return 0;
}
}  // namespace
}  // namespace wb2
