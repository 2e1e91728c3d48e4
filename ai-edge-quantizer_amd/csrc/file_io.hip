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
//
// Round 4: a ring per DIRECTION (PCIe carries both at once) and a transfer thread per direction behind
// mi355q_file_io_submit_upload / _submit_download: the caller -- the thread that walks a model's ops and launches
// the kernels -- no longer stands still while a file is read (0.62 s of a 3.6 s Gemma-2B GPTQ run, 30 of the 60 ms of
// a 1.4 GB file -> file run); it meets an upload again at mi355q_file_io_wait, where its copies are ENQUEUED, and
// orders its consumer with an event it records itself. (Measured the hard way: an event recorded on the copy
// stream by the transfer thread and awaited -- hipStreamWaitEvent, then hipEventDestroy -- by the caller's thread
// while still pending let consumers run ahead of the copies in 1 of 3 runs; the download thread, for the same
// reason, waits for a payload's producer with hipEventSynchronize instead of ordering its stream behind it.)
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <string>
#include <unordered_map>
#include <new>
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
constexpr int kThreads = 4;            // readers (3 - 6 threads read 44 - 51 GB/s out of the page cache; 8 and more 32: tools/io_ring_bench.py)
constexpr int kWriteThreads = 4;       // writers of their own: a 1 MiB pwrite() into a fresh file holds its thread ~5 x as long as a
                                       // pread() of a cached one (page allocation), and in one pool 186 MB of writes took a third of
                                       // the thread time that 1.4 GB of reads needed (round 4: 46 -> 41 ms for a 1.4 GB file -> file run;
                                       // 2 / 4 / 6 writers: the same there within noise, a 1 GB container written after a short
                                       // quantization phase 0.30 / 0.22 / 0.23 s)

struct Latch {                       // the parts of one slot still in flight
  std::mutex m;
  std::condition_variable cv;
  int open = 0;
  int error = 0;                     // first errno of a failed or short READ into this slot (-1: unexpected end of file)
  void wait() {
    std::unique_lock<std::mutex> l(m);
    cv.wait(l, [&] { return open == 0; });
  }
  int take_error() {                 // (after wait())
    std::lock_guard<std::mutex> l(m);
    const int e = error;
    error = 0;
    return e;
  }
};

struct Job {
  int fd;
  long long at;
  unsigned char* p;
  size_t n;
  bool write;
  Latch* latch;
  unsigned char* mapped = nullptr;   // a write whose destination is memory (the output file's own shared mapping): copied, not pwritten
};

struct Pool {
  std::mutex m;
  std::condition_variable cv;
  std::deque<Job> q[2];              // [0] reads, [1] writes: each has threads of its own
  std::vector<std::thread> threads;
  bool stop = false;
  int write_error = 0;               // first errno of a failed or short WRITE; reported (and cleared) by mi355q_file_io_finish only.
                                     // Read errors stay with the slot they spoiled (Latch::error): the transfer that owns the slot
                                     // sees them before the slot's bytes are copied anywhere.

  void run(int side) {
    for (;;) {
      Job j;
      {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return stop || !q[side].empty(); });
        if (q[side].empty()) return;
        j = q[side].front();
        q[side].pop_front();
      }
      size_t done = 0;
      int err = 0;
      if (j.mapped != nullptr) {       // (pages that exist: no inode lock, the copies of several threads run side by side)
        memcpy(j.mapped, j.p, j.n);
        done = j.n;
      }
      while (done < j.n) {
        const ssize_t k = j.write ? pwrite(j.fd, j.p + done, j.n - done, j.at + static_cast<long long>(done))
                                  : pread(j.fd, j.p + done, j.n - done, j.at + static_cast<long long>(done));
        if (k < 0 && errno == EINTR) continue;
        if (k <= 0) { err = k < 0 ? errno : -1; break; }
        done += static_cast<size_t>(k);
      }
      if (err && j.write) {
        std::lock_guard<std::mutex> l(m);
        if (!write_error) write_error = err;
      }
      {
        std::lock_guard<std::mutex> l(j.latch->m);
        if (err && !j.write && !j.latch->error) j.latch->error = err;
        if (--j.latch->open == 0) j.latch->cv.notify_all();
      }
    }
  }
  void start() {
    if (!threads.empty()) return;
    const char* e = getenv("MI355Q_IO_THREADS");
    const int n = e && atoi(e) > 0 && atoi(e) <= 64 ? atoi(e) : kThreads;
    const char* w = getenv("MI355Q_IO_WRITE_THREADS");
    const int nw = w && atoi(w) > 0 && atoi(w) <= 64 ? atoi(w) : kWriteThreads;
    for (int i = 0; i < n; ++i) threads.emplace_back([this] { run(0); });
    for (int i = 0; i < nw; ++i) threads.emplace_back([this] { run(1); });
  }
  void submit(int fd, long long at, unsigned char* p, size_t n, bool write, Latch* latch, unsigned char* mapped = nullptr) {
    const int parts = static_cast<int>((n + kPartBytes - 1) / kPartBytes);
    {
      std::lock_guard<std::mutex> l(latch->m);
      latch->open += parts;
    }
    {
      std::lock_guard<std::mutex> l(m);
      for (size_t o = 0; o < n; o += kPartBytes)
        q[write ? 1 : 0].push_back(Job{fd, at + static_cast<long long>(o), p + o, n - o < kPartBytes ? n - o : kPartBytes, write, latch,
                                       mapped ? mapped + o : nullptr});
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
    write_error = 0;
  }
};

struct Ring {
  unsigned char* pinned[kSlots] = {nullptr, nullptr, nullptr};
  hipEvent_t left[kSlots] = {nullptr, nullptr, nullptr};   // the slot's last device copy is done
  Latch io[kSlots];                                        // the slot's reads / writes are done
  int next = 0;
  bool ready = false;
};

std::mutex g_mutex[2];       // one transfer per direction enqueues at a time (a ring per device and direction -- PCIe carries both
                             // ways at once --, the pool per process); [0] uploads, [1] downloads
Pool* g_pool = nullptr;      // owned by the process that started its threads (g_pool_pid)
pid_t g_pool_pid = 0;
Ring g_ring[2][64];

// The io threads exist only in the process that started them: a fork()ed child inherits the Pool object with a
// non-empty thread list, queue and (possibly held) mutexes but none of the threads, and every transfer would wait on a
// latch nobody opens. The child therefore abandons the inherited pool (never destroyed: its std::thread objects are
// joinable and name threads of another process) and starts its own. (The HIP runtime does not survive fork() either;
// the ring's pinned slots are re-made the same way.)
std::mutex g_setup_mutex;    // pool, io threads and rings are made on first use, from either direction's thread

Pool& pool() {
  std::lock_guard<std::mutex> setup(g_setup_mutex);
  const pid_t me = getpid();
  if (!g_pool || g_pool_pid != me) {
    if (g_pool)
      for (auto& side : g_ring)
        for (Ring& r : side) new (&r) Ring();
    g_pool = new Pool();
    g_pool_pid = me;
  }
  return *g_pool;
}

void free_ring(Ring& r) {
  for (int s = 0; s < kSlots; ++s) {
    if (r.left[s]) (void)hipEventDestroy(r.left[s]);
    if (r.pinned[s]) (void)hipHostFree(r.pinned[s]);
    r.pinned[s] = nullptr;
    r.left[s] = nullptr;
  }
  r.ready = false;
  r.next = 0;
}

Ring* ring_of_current_device(int side) {
  Pool& p = pool();
  std::lock_guard<std::mutex> setup(g_setup_mutex);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  Ring& r = g_ring[side][dev];
  if (!r.ready) {
    for (int s = 0; s < kSlots; ++s) {
      if (hipHostMalloc(reinterpret_cast<void**>(&r.pinned[s]), kSlotBytes, hipHostMallocDefault) != hipSuccess ||
          hipEventCreateWithFlags(&r.left[s], hipEventDisableTiming) != hipSuccess) {
        const hipError_t why = hipGetLastError();
        free_ring(r);                  // a half-made ring is given back, not left pinned
        (void)why;
        return nullptr;
      }
    }
    r.ready = true;
  }
  p.start();
  return &r;
}

int32_t describe_io_error(const char* what, int err) {
  if (!err) return MI355Q_OK;
  return fail(MI355Q_IO_ERROR, "%s: %s", what, err < 0 ? "unexpected end of file" : strerror(err));
}

}  // namespace

void stop_transfer_driver();     // (below: the thread behind mi355q_file_io_submit_*)

void release_file_io() {
  stop_transfer_driver();
  std::lock_guard<std::mutex> up(g_mutex[0]);
  std::lock_guard<std::mutex> down(g_mutex[1]);
  if (!g_pool || g_pool_pid != getpid()) return;       // nothing of this process to give back
  for (auto& side : g_ring)
    for (Ring& r : side) {
      if (!r.ready) continue;
      for (int s = 0; s < kSlots; ++s) {
        r.io[s].wait();
        if (r.left[s]) (void)hipEventSynchronize(r.left[s]);
      }
      free_ring(r);
    }
  g_pool->shutdown();
}

}  // namespace mi355q

using namespace mi355q;

namespace {

// gate: an event of the caller's that `copy_stream` waits for before the first copy (a download's producer), or null
int32_t upload_body(int32_t fd, int64_t file_offset, int64_t nbytes, void* dst, void* copy_stream) {
  if (nbytes < 0 || file_offset < 0 || fd < 0) return fail(MI355Q_BAD_ARG, "bad file range");
  if (nbytes == 0) return MI355Q_OK;
  if (!dst) return fail(MI355Q_BAD_ARG, "null pointer");
  std::lock_guard<std::mutex> lock(g_mutex[0]);
  Ring* r = ring_of_current_device(0);
  if (!r) return fail(MI355Q_HIP_ERROR, "pinned staging for the io ring: %s", hipGetErrorString(hipGetLastError()));
  hipStream_t st = as_stream(copy_stream);
  Pool& p = pool();
  struct Sent { int slot; long long off; size_t size; };
  std::deque<Sent> reading;
  int read_error = 0;                 // once set, no further slot is copied to the device (its bytes are not the file's)
  auto send_oldest = [&]() -> hipError_t {
    const Sent s = reading.front();
    reading.pop_front();
    r->io[s.slot].wait();
    if (const int e = r->io[s.slot].take_error())
      if (!read_error) read_error = e;
    if (read_error) return hipSuccess;
    if (hipError_t e = hipMemcpyAsync(static_cast<unsigned char*>(dst) + s.off, r->pinned[s.slot], s.size, hipMemcpyHostToDevice, st)) return e;
    return hipEventRecord(r->left[s.slot], st);
  };
  hipError_t hip_error = hipSuccess;   // (slots still being read are always waited for: the io threads write into them)
  for (long long off = 0; off < nbytes && !read_error && !hip_error; off += static_cast<long long>(kSlotBytes)) {
    const size_t size = static_cast<size_t>(nbytes - off < static_cast<long long>(kSlotBytes) ? nbytes - off : static_cast<long long>(kSlotBytes));
    if (static_cast<int>(reading.size()) >= kSlots - 1) {
      hip_error = send_oldest();
      if (read_error || hip_error) break;
    }
    const int slot = r->next;
    r->next = (r->next + 1) % kSlots;
    r->io[slot].wait();                                  // (writes of an earlier download)
    (void)hipEventSynchronize(r->left[slot]);            // the slot's previous copy has left it
    p.submit(fd, file_offset + off, r->pinned[slot], size, false, &r->io[slot]);
    reading.push_back(Sent{slot, off, size});
  }
  while (!reading.empty()) {
    const hipError_t e = send_oldest();
    if (e && !hip_error) hip_error = e;
  }
  if (read_error) return describe_io_error("reading the model file", read_error);
  if (hip_error) return fail(MI355Q_HIP_ERROR, "upload copy: %s", hipGetErrorString(hip_error));
  return MI355Q_OK;
}

int32_t download_body(const void* src, int64_t nbytes, int32_t fd, int64_t file_offset, void* copy_stream, hipEvent_t gate,
                      unsigned char* mapped = nullptr) {
  if (nbytes < 0 || file_offset < 0 || (fd < 0 && !mapped)) return fail(MI355Q_BAD_ARG, "bad file range");
  if (nbytes == 0) return MI355Q_OK;
  if (!src) return fail(MI355Q_BAD_ARG, "null pointer");
  std::lock_guard<std::mutex> lock(g_mutex[1]);
  Ring* r = ring_of_current_device(1);
  if (!r) return fail(MI355Q_HIP_ERROR, "pinned staging for the io ring: %s", hipGetErrorString(hipGetLastError()));
  hipStream_t st = as_stream(copy_stream);
  // (the payload's producer has finished before the first copy is enqueued: this thread's only job is to wait, and the
  // copy stream then never holds a copy that waits behind compute)
  if (gate)
    if (hipError_t e = hipEventSynchronize(gate)) return fail(MI355Q_HIP_ERROR, "download gate: %s", hipGetErrorString(e));
  Pool& p = pool();
  int pending_slot = -1;
  long long pending_off = 0;
  size_t pending_size = 0;
  auto drain = [&] {            // the pending slot's copy has arrived: hand it to the writers
    (void)hipEventSynchronize(r->left[pending_slot]);
    p.submit(fd, file_offset + pending_off, r->pinned[pending_slot], pending_size, true, &r->io[pending_slot],
             mapped ? mapped + pending_off : nullptr);
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

// ---- transfers that do not hold the caller: mi355q_file_io_submit_upload / _download / mi355q_file_io_wait ------------
// The two bodies above return when their last device copy is ENQUEUED, which for an upload is when the last pread is
// done: a model's weights cross at the ring's rate while the calling thread stands still. Submitted transfers run the
// same bodies on one driver thread, first in first out; the caller meets a transfer again only where it needs it
// (mi355q_file_io_wait: its device copies are enqueued and `consumer_stream` is ordered behind them).
struct Transfer {
  bool upload = true;
  int32_t fd = -1;
  int64_t file_offset = 0, nbytes = 0;
  void* device_ptr = nullptr;
  void* stream = nullptr;
  hipEvent_t gate = nullptr;      // download: the caller's event behind which the payload is final (may be null)
  unsigned char* mapped = nullptr; // download: destination in memory instead of (fd, file_offset)
  int device = 0;
  bool enqueued = false;
  int32_t status = MI355Q_OK;
  std::string message;
};

struct Driver {
  std::mutex m;
  std::condition_variable cv;          // work for the thread / a transfer has been enqueued
  std::deque<int64_t> queue;
  std::unordered_map<int64_t, Transfer> all;
  int64_t next_ticket = 1;
  bool busy = false, stop = false;
  int32_t download_status = MI355Q_OK;   // first failure of a submitted download: reported (and cleared) by mi355q_file_io_finish
  std::string download_message;
  std::thread thread;
  pid_t pid = 0;

  void run() {
    for (;;) {
      int64_t ticket;
      Transfer t;
      {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return stop || !queue.empty(); });
        if (queue.empty()) return;
        ticket = queue.front();
        queue.pop_front();
        t = all[ticket];
        busy = true;
      }
      int32_t status = MI355Q_OK;
      std::string message;
      if (hipSetDevice(t.device) != hipSuccess) {
        status = MI355Q_HIP_ERROR;
        message = "hipSetDevice on the transfer thread failed";
      } else {
        status = t.upload ? upload_body(t.fd, t.file_offset, t.nbytes, t.device_ptr, t.stream)
                          : download_body(t.device_ptr, t.nbytes, t.fd, t.file_offset, t.stream, t.gate, t.mapped);
        if (status != MI355Q_OK) message = mi355q_last_error();      // (this thread's)
      }
      {
        std::lock_guard<std::mutex> l(m);
        auto it = all.find(ticket);
        if (it != all.end()) {
          if (t.upload) {
            it->second.enqueued = true;
            it->second.status = status;
            it->second.message = message;
          } else {                         // nobody waits for a download: its record goes, its failure stays
            all.erase(it);
            if (status != MI355Q_OK && download_status == MI355Q_OK) {
              download_status = status;
              download_message = message;
            }
          }
        }
        busy = false;
      }
      cv.notify_all();
    }
  }
};

Driver* g_driver[2] = {nullptr, nullptr};     // [0] uploads, [1] downloads: the two directions do not queue behind each other
std::mutex g_driver_mutex;

Driver* driver_if_any(int side) {
  std::lock_guard<std::mutex> l(g_driver_mutex);
  Driver* d = g_driver[side];
  return d && d->pid == getpid() ? d : nullptr;
}

Driver& driver(int side) {
  std::lock_guard<std::mutex> l(g_driver_mutex);
  const pid_t me = getpid();
  if (!g_driver[side] || g_driver[side]->pid != me) {     // (a fork()ed child has the object but not the thread: it starts its own)
    g_driver[side] = new Driver();
    g_driver[side]->pid = me;
    g_driver[side]->thread = std::thread([d = g_driver[side]] { d->run(); });
  }
  return *g_driver[side];
}

int32_t submit(Transfer t, int64_t* ticket) {
  if (hipGetDevice(&t.device) != hipSuccess) return fail(MI355Q_HIP_ERROR, "hipGetDevice: %s", hipGetErrorString(hipGetLastError()));
  Driver& d = driver(t.upload ? 0 : 1);
  {
    std::lock_guard<std::mutex> l(d.m);
    *ticket = d.next_ticket++;
    d.all[*ticket] = t;
    d.queue.push_back(*ticket);
  }
  d.cv.notify_all();
  return MI355Q_OK;
}

}  // namespace

namespace mi355q {
void stop_transfer_driver() {
  for (int side = 0; side < 2; ++side) {
    Driver* d = driver_if_any(side);
    if (!d) continue;
    {
      std::lock_guard<std::mutex> l(d->m);
      d->stop = true;
    }
    d->cv.notify_all();
    if (d->thread.joinable()) d->thread.join();        // (runs the queue dry first)
    std::lock_guard<std::mutex> l(g_driver_mutex);
    delete d;
    g_driver[side] = nullptr;
  }
}
}  // namespace mi355q

extern "C" int32_t mi355q_file_to_device(int32_t fd, int64_t file_offset, int64_t nbytes, void* dst, void* copy_stream) {
  clear_error();
  return upload_body(fd, file_offset, nbytes, dst, copy_stream);
}

extern "C" int32_t mi355q_device_to_file(const void* src, int64_t nbytes, int32_t fd, int64_t file_offset, void* copy_stream) {
  clear_error();
  return download_body(src, nbytes, fd, file_offset, copy_stream, nullptr);
}

extern "C" int32_t mi355q_file_io_submit_upload(int32_t fd, int64_t file_offset, int64_t nbytes, void* dst, void* copy_stream,
                                                int64_t* ticket) {
  clear_error();
  if (!ticket) return fail(MI355Q_BAD_ARG, "null ticket");
  if (nbytes < 0 || file_offset < 0 || fd < 0) return fail(MI355Q_BAD_ARG, "bad file range");
  if (nbytes > 0 && !dst) return fail(MI355Q_BAD_ARG, "null pointer");
  Transfer t;
  t.upload = true; t.fd = fd; t.file_offset = file_offset; t.nbytes = nbytes; t.device_ptr = dst; t.stream = copy_stream;
  return submit(t, ticket);
}

extern "C" int32_t mi355q_file_io_submit_download(const void* src, int64_t nbytes, int32_t fd, int64_t file_offset, void* copy_stream,
                                                  void* ready_event) {
  clear_error();
  if (nbytes < 0 || file_offset < 0 || fd < 0) return fail(MI355Q_BAD_ARG, "bad file range");
  if (nbytes == 0) return MI355Q_OK;
  if (!src) return fail(MI355Q_BAD_ARG, "null pointer");
  Transfer t;
  t.upload = false; t.fd = fd; t.file_offset = file_offset; t.nbytes = nbytes; t.device_ptr = const_cast<void*>(src); t.stream = copy_stream;
  t.gate = reinterpret_cast<hipEvent_t>(ready_event);
  int64_t ticket = 0;
  return submit(t, &ticket);
}

extern "C" int32_t mi355q_file_io_submit_download_mapped(const void* src, int64_t nbytes, void* dst, void* copy_stream,
                                                         void* ready_event) {
  clear_error();
  if (nbytes < 0) return fail(MI355Q_BAD_ARG, "negative size");
  if (nbytes == 0) return MI355Q_OK;
  if (!src || !dst) return fail(MI355Q_BAD_ARG, "null pointer");
  Transfer t;
  t.upload = false; t.fd = -1; t.file_offset = 0; t.nbytes = nbytes; t.device_ptr = const_cast<void*>(src); t.stream = copy_stream;
  t.gate = reinterpret_cast<hipEvent_t>(ready_event);
  t.mapped = static_cast<unsigned char*>(dst);
  int64_t ticket = 0;
  return submit(t, &ticket);
}

extern "C" int32_t mi355q_file_io_wait(int64_t ticket) {
  clear_error();
  Driver* d = driver_if_any(0);
  if (!d) return fail(MI355Q_BAD_ARG, "no such transfer (ticket %lld)", static_cast<long long>(ticket));
  Transfer t;
  {
    std::unique_lock<std::mutex> l(d->m);
    auto it = d->all.find(ticket);
    if (it == d->all.end()) return fail(MI355Q_BAD_ARG, "no such transfer (ticket %lld)", static_cast<long long>(ticket));
    d->cv.wait(l, [&] { return d->all[ticket].enqueued; });
    t = d->all[ticket];
    d->all.erase(ticket);
  }
  // (the caller orders its consumers behind the copies the way it always did: an event it records on the copy stream NOW
  // lies behind every copy of this transfer, all of which are enqueued)
  if (t.status != MI355Q_OK) return fail(static_cast<mi355q_status>(t.status), "%s", t.message.c_str());
  return MI355Q_OK;
}

extern "C" int32_t mi355q_file_io_finish(void) {
  clear_error();
  int32_t submitted_status = MI355Q_OK;
  std::string submitted_message;
  for (int side = 0; side < 2; ++side) {      // submitted transfers first: all enqueued, none running
    Driver* d = driver_if_any(side);
    if (!d) continue;
    std::unique_lock<std::mutex> l(d->m);
    d->cv.wait(l, [&] { return d->queue.empty() && !d->busy; });
    if (d->download_status != MI355Q_OK && submitted_status == MI355Q_OK) {
      submitted_status = d->download_status;
      submitted_message = d->download_message;
    }
    d->download_status = MI355Q_OK;
    d->download_message.clear();
  }
  std::lock_guard<std::mutex> up(g_mutex[0]);
  std::lock_guard<std::mutex> down(g_mutex[1]);
  if (!g_pool || g_pool_pid != getpid()) {                   // no ring transfer of this process is open ...
    // ... but a submitted one may have failed before it reached the pool (hipSetDevice on the driver thread): its status
    // was taken (and cleared) above and must not be lost here
    if (submitted_status != MI355Q_OK) return fail(static_cast<mi355q_status>(submitted_status), "%s", submitted_message.c_str());
    return MI355Q_OK;
  }
  for (auto& side : g_ring)
    for (Ring& r : side)
      if (r.ready)
        for (int s = 0; s < kSlots; ++s) r.io[s].wait();
  int err;
  {
    std::lock_guard<std::mutex> l(g_pool->m);
    err = g_pool->write_error;
    g_pool->write_error = 0;
  }
  if (submitted_status != MI355Q_OK) return fail(static_cast<mi355q_status>(submitted_status), "%s", submitted_message.c_str());
  return describe_io_error("writing the output file", err);
}
