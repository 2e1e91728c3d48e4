// Model files <-> HBM: the io ring of the file path (host code only; no kernels).
//
//   ref: utils/tfl_flatbuffer_utils.py:142-163 (the model file is mapped / read whole), model_modifier.py:290-391
//        (the serializers copy every quantized buffer into one host bytearray and write it out)
//
// The reference touches every weight byte on the host; here a weight crosses the host only as
// bytes in flight. A pageable hipMemcpy of a view of the mapped file moves 24 GB/s (one staging
// thread inside the runtime, and every page of the mapping is faulted in first); a pageable copy
// into the mapping of a fresh output file 10 GB/s. The ring: three pinned slots of 8 MiB per
// device (page-locking costs ~1 ms per MiB, once per process), four io threads that pread()
// the file into a slot / pwrite() a slot to the file in 1 MiB parts (kernel copies from / to the
// page cache: no page tables of a mapping are populated), and asynchronous copies on the caller's
// copy stream, so that the reads of the next slots run while a slot's copy is in flight.
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include <errno.h>
#include <string.h>
#include <unistd.h>

#include "common.h"

namespace mi355q {
namespace {

constexpr size_t kSlotBytes = 8u << 20;
constexpr int kSlots = 3;
constexpr size_t kPartBytes = 1u << 20;
constexpr int kThreads = 4;            // (3 - 6 threads read 44 - 51 GB/s out of the page cache; 8 and more 32: tools/io_ring_bench.py)

struct Latch {                       // the parts of one slot still in flight
  std::mutex m;
  std::condition_variable cv;
  int open = 0;
  void wait() {
    std::unique_lock<std::mutex> l(m);
    cv.wait(l, [&] { return open == 0; });
  }
};

struct Job {
  int fd;
  long long at;
  unsigned char* p;
  size_t n;
  bool write;
  Latch* latch;
};

struct Pool {
  std::mutex m;
  std::condition_variable cv;
  std::deque<Job> q;
  std::vector<std::thread> threads;
  bool stop = false;
  int error = 0;                     // first errno of a failed or short transfer (-1: unexpected end of file)

  void run() {
    for (;;) {
      Job j;
      {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return stop || !q.empty(); });
        if (q.empty()) return;
        j = q.front();
        q.pop_front();
      }
      size_t done = 0;
      int err = 0;
      while (done < j.n) {
        const ssize_t k = j.write ? pwrite(j.fd, j.p + done, j.n - done, j.at + static_cast<long long>(done))
                                  : pread(j.fd, j.p + done, j.n - done, j.at + static_cast<long long>(done));
        if (k < 0 && errno == EINTR) continue;
        if (k <= 0) { err = k < 0 ? errno : -1; break; }
        done += static_cast<size_t>(k);
      }
      if (err) {
        std::lock_guard<std::mutex> l(m);
        if (!error) error = err;
      }
      {
        std::lock_guard<std::mutex> l(j.latch->m);
        if (--j.latch->open == 0) j.latch->cv.notify_all();
      }
    }
  }
  void start() {
    if (!threads.empty()) return;
    const char* e = getenv("MI355Q_IO_THREADS");
    const int n = e && atoi(e) > 0 && atoi(e) <= 64 ? atoi(e) : kThreads;
    for (int i = 0; i < n; ++i) threads.emplace_back([this] { run(); });
  }
  void submit(int fd, long long at, unsigned char* p, size_t n, bool write, Latch* latch) {
    const int parts = static_cast<int>((n + kPartBytes - 1) / kPartBytes);
    {
      std::lock_guard<std::mutex> l(latch->m);
      latch->open += parts;
    }
    {
      std::lock_guard<std::mutex> l(m);
      for (size_t o = 0; o < n; o += kPartBytes)
        q.push_back(Job{fd, at + static_cast<long long>(o), p + o, n - o < kPartBytes ? n - o : kPartBytes, write, latch});
    }
    cv.notify_all();
  }
  ~Pool() { shutdown(); }            // (joinable threads at process exit would call std::terminate)
  void shutdown() {
    {
      std::lock_guard<std::mutex> l(m);
      stop = true;
    }
    cv.notify_all();
    for (std::thread& t : threads) t.join();
    threads.clear();
    stop = false;
    error = 0;
  }
};

struct Ring {
  unsigned char* pinned[kSlots] = {nullptr, nullptr, nullptr};
  hipEvent_t left[kSlots] = {nullptr, nullptr, nullptr};   // the slot's last device copy is done
  Latch io[kSlots];                                        // the slot's reads / writes are done
  int next = 0;
  bool ready = false;
};

std::mutex g_mutex;          // one transfer at a time enqueues (the ring is per device, the pool per process)
Pool g_pool;
Ring g_ring[64];

Ring* ring_of_current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  Ring& r = g_ring[dev];
  if (!r.ready) {
    for (int s = 0; s < kSlots; ++s) {
      if (hipHostMalloc(reinterpret_cast<void**>(&r.pinned[s]), kSlotBytes, hipHostMallocDefault) != hipSuccess) return nullptr;
      if (hipEventCreateWithFlags(&r.left[s], hipEventDisableTiming) != hipSuccess) return nullptr;
    }
    r.ready = true;
  }
  g_pool.start();
  return &r;
}

int32_t io_error(const char* what) {
  int err;
  {
    std::lock_guard<std::mutex> l(g_pool.m);
    err = g_pool.error;
    g_pool.error = 0;
  }
  if (!err) return MI355Q_OK;
  return fail(MI355Q_IO_ERROR, "%s: %s", what, err < 0 ? "unexpected end of file" : strerror(err));
}

}  // namespace

void release_file_io() {
  std::lock_guard<std::mutex> lock(g_mutex);
  for (Ring& r : g_ring) {
    if (!r.ready) continue;
    for (int s = 0; s < kSlots; ++s) {
      r.io[s].wait();
      if (r.left[s]) { (void)hipEventSynchronize(r.left[s]); (void)hipEventDestroy(r.left[s]); }
      if (r.pinned[s]) (void)hipHostFree(r.pinned[s]);
      r.pinned[s] = nullptr;
      r.left[s] = nullptr;
    }
    r.ready = false;
    r.next = 0;
  }
  g_pool.shutdown();
}

}  // namespace mi355q

using namespace mi355q;

extern "C" int32_t mi355q_file_to_device(int32_t fd, int64_t file_offset, int64_t nbytes, void* dst, void* copy_stream) {
  clear_error();
  if (nbytes < 0 || file_offset < 0 || fd < 0) return fail(MI355Q_BAD_ARG, "bad file range");
  if (nbytes == 0) return MI355Q_OK;
  if (!dst) return fail(MI355Q_BAD_ARG, "null pointer");
  std::lock_guard<std::mutex> lock(g_mutex);
  Ring* r = ring_of_current_device();
  if (!r) return fail(MI355Q_HIP_ERROR, "pinned staging for the io ring: %s", hipGetErrorString(hipGetLastError()));
  hipStream_t st = as_stream(copy_stream);
  struct Sent { int slot; long long off; size_t size; };
  std::deque<Sent> reading;
  auto send_oldest = [&]() -> hipError_t {
    const Sent s = reading.front();
    reading.pop_front();
    r->io[s.slot].wait();
    if (hipError_t e = hipMemcpyAsync(static_cast<unsigned char*>(dst) + s.off, r->pinned[s.slot], s.size, hipMemcpyHostToDevice, st)) return e;
    return hipEventRecord(r->left[s.slot], st);
  };
  for (long long off = 0; off < nbytes; off += static_cast<long long>(kSlotBytes)) {
    const size_t size = static_cast<size_t>(nbytes - off < static_cast<long long>(kSlotBytes) ? nbytes - off : static_cast<long long>(kSlotBytes));
    if (static_cast<int>(reading.size()) >= kSlots - 1)
      if (hipError_t e = send_oldest()) return fail(MI355Q_HIP_ERROR, "upload copy: %s", hipGetErrorString(e));
    const int slot = r->next;
    r->next = (r->next + 1) % kSlots;
    r->io[slot].wait();                                  // (writes of an earlier download)
    (void)hipEventSynchronize(r->left[slot]);            // the slot's previous copy has left it
    g_pool.submit(fd, file_offset + off, r->pinned[slot], size, false, &r->io[slot]);
    reading.push_back(Sent{slot, off, size});
  }
  while (!reading.empty())
    if (hipError_t e = send_oldest()) return fail(MI355Q_HIP_ERROR, "upload copy: %s", hipGetErrorString(e));
  return io_error("reading the model file");
}

extern "C" int32_t mi355q_device_to_file(const void* src, int64_t nbytes, int32_t fd, int64_t file_offset, void* copy_stream) {
  clear_error();
  if (nbytes < 0 || file_offset < 0 || fd < 0) return fail(MI355Q_BAD_ARG, "bad file range");
  if (nbytes == 0) return MI355Q_OK;
  if (!src) return fail(MI355Q_BAD_ARG, "null pointer");
  std::lock_guard<std::mutex> lock(g_mutex);
  Ring* r = ring_of_current_device();
  if (!r) return fail(MI355Q_HIP_ERROR, "pinned staging for the io ring: %s", hipGetErrorString(hipGetLastError()));
  hipStream_t st = as_stream(copy_stream);
  int pending_slot = -1;
  long long pending_off = 0;
  size_t pending_size = 0;
  auto drain = [&] {            // the pending slot's copy has arrived: hand it to the writers
    (void)hipEventSynchronize(r->left[pending_slot]);
    g_pool.submit(fd, file_offset + pending_off, r->pinned[pending_slot], pending_size, true, &r->io[pending_slot]);
  };
  for (long long off = 0; off < nbytes; off += static_cast<long long>(kSlotBytes)) {
    const size_t size = static_cast<size_t>(nbytes - off < static_cast<long long>(kSlotBytes) ? nbytes - off : static_cast<long long>(kSlotBytes));
    const int slot = r->next;
    r->next = (r->next + 1) % kSlots;
    r->io[slot].wait();
    (void)hipEventSynchronize(r->left[slot]);
    if (hipError_t e = hipMemcpyAsync(r->pinned[slot], static_cast<const unsigned char*>(src) + off, size, hipMemcpyDeviceToHost, st))
      return fail(MI355Q_HIP_ERROR, "download copy: %s", hipGetErrorString(e));
    (void)hipEventRecord(r->left[slot], st);
    if (pending_slot >= 0) drain();
    pending_slot = slot; pending_off = off; pending_size = size;
  }
  if (pending_slot >= 0) drain();
  return MI355Q_OK;               // (write errors surface in mi355q_file_io_finish)
}

extern "C" int32_t mi355q_file_io_finish(void) {
  clear_error();
  {
    std::lock_guard<std::mutex> lock(g_mutex);
    for (Ring& r : g_ring)
      if (r.ready)
        for (int s = 0; s < kSlots; ++s) r.io[s].wait();
  }
  return io_error("writing the output file");
}
