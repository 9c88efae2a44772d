// host_staging.cpp — pinned slab, kept device scratch and the piecewise host <-> device moves (host_staging.h).
#include "host_staging.h"
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <thread>

namespace lzf_host {

namespace {
constexpr size_t kPiece = 4u << 20;          // bytes per memcpy task / DMA: small enough to pipeline, large enough for full PCIe rate
constexpr size_t kInline = 256u << 10;       // moves this small are done by the calling thread
inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// One task = consecutive pieces of segments whose slab range [a, b) is moved by one DMA.
struct Piece { size_t slab_off; uint8_t* host; size_t len; };
struct Task { size_t a, b; size_t first, count; };
void plan(const std::vector<Seg>& segs, std::vector<Piece>& pieces, std::vector<Task>& tasks) {
    for (const Seg& s : segs)
        for (size_t o = 0; o < s.len; o += kPiece) pieces.push_back({s.slab_off + o, s.host + o, s.len - o < kPiece ? s.len - o : kPiece});
    for (size_t i = 0; i < pieces.size();) {
        Task t{pieces[i].slab_off, pieces[i].slab_off + pieces[i].len, i, 1};
        size_t j = i + 1;
        while (j < pieces.size() && pieces[j].slab_off >= t.b && pieces[j].slab_off + pieces[j].len - t.a <= kPiece) { t.b = pieces[j].slab_off + pieces[j].len; ++t.count; ++j; }
        tasks.push_back(t);
        i = j;
    }
}
}  // namespace

struct Staging::Pool {
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv, cv_done;
    std::deque<std::function<void()>> q;
    size_t pending = 0;
    bool stop = false;
    explicit Pool(unsigned n) {
        for (unsigned i = 0; i < n; ++i) th.emplace_back([this] { run(); });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> g(m); stop = true; }
        cv.notify_all();
        for (auto& t : th) t.join();
    }
    void run() {
        for (;;) {
            std::function<void()> f;
            { std::unique_lock<std::mutex> g(m);
              cv.wait(g, [this] { return stop || !q.empty(); });
              if (q.empty()) return;
              f = std::move(q.front()); q.pop_front(); }
            f();
            { std::lock_guard<std::mutex> g(m); --pending; }
            cv_done.notify_all();
        }
    }
    void submit(std::function<void()> f) {
        { std::lock_guard<std::mutex> g(m); q.push_back(std::move(f)); ++pending; }
        cv.notify_one();
    }
    // the calling thread works the queue down too, then waits for the tasks still running
    void wait() {
        for (;;) {
            std::function<void()> f;
            { std::lock_guard<std::mutex> g(m); if (!q.empty()) { f = std::move(q.front()); q.pop_front(); } }
            if (!f) break;
            f();
            { std::lock_guard<std::mutex> g(m); --pending; }
        }
        std::unique_lock<std::mutex> g(m);
        cv_done.wait(g, [this] { return pending == 0; });
    }
    // wait for a condition a task establishes.  With no worker threads at all — lzf_frame_set_host_threads(LZF_HOST_THREADS_NONE) — the
    // calling thread runs the queued tasks itself, in order; with workers it only waits: it is the thread that issues the DMA of a
    // finished piece, and inside a 4 MiB memcpy of its own it would issue that DMA late (the staging would serialise with the link)
    template <class Pred> void wait_until(Pred p) {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> g(m);
                if (p()) return;
                if (!th.empty()) { cv_done.wait(g, [&] { return p(); }); return; }
                if (q.empty()) { cv_done.wait(g, [&] { return p() || !q.empty(); }); if (p()) return; }
                if (!q.empty()) { f = std::move(q.front()); q.pop_front(); }
            }
            if (f) {
                f();
                { std::lock_guard<std::mutex> g(m); --pending; }
                cv_done.notify_all();
            }
        }
    }
};

// One instance per device (the calling thread's current device picks it), created on first use and never destroyed (no HIP
// calls at exit): a process that drives several GPUs keeps a slab and scratch on each.
Staging& Staging::get() {
    static std::mutex m;
    static Staging* inst[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); dev = 0; }
    std::lock_guard<std::mutex> g(m);
    if (!inst[dev]) inst[dev] = new Staging;
    return *inst[dev];
}

Staging::Pool* Staging::pool() {
    unsigned want = want_threads_;
    if (!want) { const unsigned hw = std::thread::hardware_concurrency(); want = hw >= 48 ? 12 : hw >= 8 ? hw / 4 : 2; }
    if (want == kNoThreads) want = 0;                  // every staging copy and hash on the calling thread
    if (pool_ && pool_->th.size() != want) { delete pool_; pool_ = nullptr; }
    if (!pool_) pool_ = new Pool(want);
    return pool_;
}
void Staging::set_threads(unsigned n) { want_threads_ = n == 0xFFFFFFFFu ? kNoThreads : n > 64 ? 64 : n; }

void Staging::release() {
    if (pin_) (void)hipHostFree(pin_);
    pin_ = nullptr; pin_cap_ = 0;
    if (mail_) (void)hipHostFree(mail_);
    mail_ = nullptr; mail_cap_ = 0;
    for (int i = 0; i < kSlots; ++i) { if (dev_[i]) (void)hipFree(dev_[i]); dev_[i] = nullptr; dev_cap_[i] = 0; }
    for (auto& s : streams_) { if (s) (void)hipStreamDestroy(s); s = nullptr; }
    for (auto e : events_) (void)hipEventDestroy(e);
    events_.clear();
    device_ = -1;
    (void)hipGetLastError();
}

uint8_t* Staging::pinned(size_t bytes) {
    int dev = -1; if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    if (device_ != dev) { release(); device_ = dev; }
    if (bytes <= pin_cap_ && pin_) return pin_;
    if (pin_) { (void)hipHostFree(pin_); pin_ = nullptr; pin_cap_ = 0; }
    // (small calls pin little: 1 MiB steps below 16 MiB, 64 MiB steps beyond)
    const size_t cap = round_up(bytes ? bytes : 1, bytes < (16u << 20) ? (1u << 20) : (64u << 20));
    void* p = nullptr;
    if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) return nullptr;
    pin_ = static_cast<uint8_t*>(p); pin_cap_ = cap;
    return pin_;
}

uint8_t* Staging::mailbox(size_t bytes) {
    int dev = -1; if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    if (device_ != dev) { release(); device_ = dev; }
    if (bytes <= mail_cap_ && mail_) return mail_;
    if (mail_) { (void)hipHostFree(mail_); mail_ = nullptr; mail_cap_ = 0; }
    const size_t cap = round_up(bytes ? bytes : 1, 1u << 20);
    void* p = nullptr;
    if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) return nullptr;
    mail_ = static_cast<uint8_t*>(p); mail_cap_ = cap;
    return mail_;
}

void* Staging::device(int slot, size_t bytes) {
    int dev = -1; if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    if (device_ != dev) { release(); device_ = dev; }
    if (bytes <= dev_cap_[slot] && dev_[slot]) return dev_[slot];
    if (dev_[slot]) { (void)hipFree(dev_[slot]); dev_[slot] = nullptr; dev_cap_[slot] = 0; }
    const size_t cap = round_up(bytes ? bytes : 1, bytes >= (16u << 20) ? (16u << 20) : 4096);
    void* p = nullptr;
    if (hipMalloc(&p, cap) != hipSuccess) return nullptr;
    dev_[slot] = p; dev_cap_[slot] = cap;
    return p;
}

hipStream_t Staging::stream(int i) {
    int dev = -1; if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    if (device_ != dev) { release(); device_ = dev; }
    if (!streams_[i] && hipStreamCreateWithFlags(&streams_[i], hipStreamNonBlocking) != hipSuccess) streams_[i] = nullptr;
    return streams_[i];
}

hipEvent_t Staging::event(size_t i) {
    while (events_.size() <= i) {
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        events_.push_back(e);
    }
    return events_[i];
}

void Staging::parallel_for(size_t n, const std::function<void(size_t)>& fn) {
    if (n <= 1) { if (n) fn(0); return; }
    Pool* p = pool();
    for (size_t i = 0; i < n; ++i) p->submit([&fn, i] { fn(i); });
    p->wait();
}

#define TRY(e) do { hipError_t e__ = (e); if (e__ != hipSuccess) return e__; } while (0)

hipError_t Staging::upload(const std::vector<Seg>& segs, size_t slab_bytes, uint8_t* d_base) {
    size_t total = 0;
    for (const Seg& s : segs) total += s.len;
    if (!total) return hipSuccess;
    uint8_t* pin = pinned(slab_bytes);
    if (!pin) return hipErrorOutOfMemory;
    hipStream_t c1 = stream(1), c2 = stream(2);
    if (!c1 || !c2) return hipErrorUnknown;
    size_t span_a = SIZE_MAX, span_b = 0;
    for (const Seg& s : segs) if (s.len) { if (s.slab_off < span_a) span_a = s.slab_off; if (s.slab_off + s.len > span_b) span_b = s.slab_off + s.len; }
    // (one copy over the whole span only while the span is mostly payload: sparse segments — small payloads in large slots —
    //  take the piecewise path, which moves what the segments cover)
    if (total <= kInline && span_b - span_a <= 2 * total) {
        size_t a = SIZE_MAX, b = 0;
        for (const Seg& s : segs) if (s.len) { memcpy(pin + s.slab_off, s.host, s.len); if (s.slab_off < a) a = s.slab_off; if (s.slab_off + s.len > b) b = s.slab_off + s.len; }
        TRY(hipMemcpyAsync(d_base + a, pin + a, b - a, hipMemcpyHostToDevice, c1));
        ++counters.h2d_copies; counters.h2d_bytes += b - a;
    } else {
        std::vector<Piece> pieces; std::vector<Task> tasks;
        plan(segs, pieces, tasks);
        Pool* p = pool();
        std::vector<std::atomic<uint8_t>> done(tasks.size());
        for (auto& d : done) d.store(0, std::memory_order_relaxed);
        for (size_t k = 0; k < tasks.size(); ++k)
            p->submit([&, k] {
                const Task& t = tasks[k];
                for (size_t i = t.first; i < t.first + t.count; ++i) memcpy(pin + pieces[i].slab_off, pieces[i].host, pieces[i].len);
                done[k].store(1, std::memory_order_release);
            });
        hipError_t err = hipSuccess;
        for (size_t k = 0; k < tasks.size(); ++k) {
            p->wait_until([&] { return done[k].load(std::memory_order_acquire) != 0; });
            if (err == hipSuccess) err = hipMemcpyAsync(d_base + tasks[k].a, pin + tasks[k].a, tasks[k].b - tasks[k].a, hipMemcpyHostToDevice, (k & 1) ? c2 : c1);
            ++counters.h2d_copies; counters.h2d_bytes += tasks[k].b - tasks[k].a;
        }
        p->wait();
        TRY(err);
    }
    return hipSuccess;
}

hipError_t Staging::join_copies(hipStream_t waiter) {
    hipStream_t c1 = stream(1), c2 = stream(2);
    hipEvent_t e1 = event(0), e2 = event(1);
    if (!c1 || !c2 || !e1 || !e2) return hipErrorUnknown;
    TRY(hipEventRecord(e1, c1)); TRY(hipEventRecord(e2, c2));
    TRY(hipStreamWaitEvent(waiter, e1, 0)); TRY(hipStreamWaitEvent(waiter, e2, 0));
    return hipSuccess;
}

hipError_t Staging::download(const std::vector<Seg>& segs, size_t slab_bytes, const uint8_t* d_base, hipStream_t before, hipEvent_t ready) {
    size_t total = 0;
    for (const Seg& s : segs) total += s.len;
    hipStream_t c1 = stream(4), c2 = stream(5);      // the way back has its own pair of streams: it overlaps uploads still in flight
    if (!c1 || !c2) return hipErrorUnknown;
    if (before) {
        hipEvent_t e0 = event(0);
        if (!e0) return hipErrorUnknown;
        TRY(hipEventRecord(e0, before)); TRY(hipStreamWaitEvent(c1, e0, 0)); TRY(hipStreamWaitEvent(c2, e0, 0));
    }
    if (ready) { TRY(hipStreamWaitEvent(c1, ready, 0)); TRY(hipStreamWaitEvent(c2, ready, 0)); }
    if (!total) { if (before) TRY(hipStreamSynchronize(before)); if (ready) TRY(hipEventSynchronize(ready)); return hipSuccess; }
    uint8_t* pin = pinned(slab_bytes);
    if (!pin) return hipErrorOutOfMemory;
    size_t span_a = SIZE_MAX, span_b = 0;
    for (const Seg& s : segs) if (s.len) { if (s.slab_off < span_a) span_a = s.slab_off; if (s.slab_off + s.len > span_b) span_b = s.slab_off + s.len; }
    if (total <= kInline && span_b - span_a <= 2 * total) {
        size_t a = SIZE_MAX, b = 0;
        for (const Seg& s : segs) if (s.len) { if (s.slab_off < a) a = s.slab_off; if (s.slab_off + s.len > b) b = s.slab_off + s.len; }
        TRY(hipMemcpyAsync(pin + a, d_base + a, b - a, hipMemcpyDeviceToHost, c1));
        ++counters.d2h_copies; counters.d2h_bytes += b - a;
        TRY(hipStreamSynchronize(c1));
        for (const Seg& s : segs) if (s.len) memcpy(s.host, pin + s.slab_off, s.len);
        return hipSuccess;
    }
    std::vector<Piece> pieces; std::vector<Task> tasks;
    plan(segs, pieces, tasks);
    Pool* p = pool();
    for (size_t k = 0; k < tasks.size(); ++k) {
        hipEvent_t e = event(2 + k);
        if (!e) return hipErrorUnknown;
        hipStream_t c = (k & 1) ? c2 : c1;
        TRY(hipMemcpyAsync(pin + tasks[k].a, d_base + tasks[k].a, tasks[k].b - tasks[k].a, hipMemcpyDeviceToHost, c));
        ++counters.d2h_copies; counters.d2h_bytes += tasks[k].b - tasks[k].a;
        TRY(hipEventRecord(e, c));
    }
    hipError_t err = hipSuccess;
    for (size_t k = 0; k < tasks.size(); ++k) {
        if (err == hipSuccess) err = hipEventSynchronize(events_[2 + k]);
        if (err != hipSuccess) break;
        p->submit([&, k] {
            const Task& t = tasks[k];
            for (size_t i = t.first; i < t.first + t.count; ++i) memcpy(pieces[i].host, pin + pieces[i].slab_off, pieces[i].len);
        });
    }
    p->wait();
    return err;
}

}  // namespace lzf_host
