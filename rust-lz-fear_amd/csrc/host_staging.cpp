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
constexpr size_t kRingMax = (size_t)2 << 30; // the slab never grows beyond this: a larger move goes through it as a RING of kPiece slots
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
void Staging::set_pinned_limit(size_t bytes) {
    if (bytes && bytes < 2 * kPiece) bytes = 2 * kPiece;
    if (bytes == ring_max_) return;
    ring_max_ = bytes;
    if (pin_) { (void)drain_slots(); (void)hipHostFree(pin_); pin_ = nullptr; pin_cap_ = 0; next_slot_ = 0; (void)hipGetLastError(); }
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
    for (auto e : slot_ev_) if (e) (void)hipEventDestroy(e);
    slot_ev_.clear(); slot_busy_.clear(); next_slot_ = 0;
    device_ = -1;
    (void)hipGetLastError();
}

uint8_t* Staging::pinned(size_t bytes) {
    int dev = -1; if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    if (device_ != dev) { release(); device_ = dev; }
    { const size_t lim = ring_max_ ? ring_max_ : kRingMax; if (bytes > lim) bytes = lim; }     // (a move larger than the slab recycles it slot by slot: upload / download)
    if (bytes <= pin_cap_ && pin_) return pin_;
    if (pin_) { if (!drain_slots()) return nullptr; (void)hipHostFree(pin_); pin_ = nullptr; pin_cap_ = 0; }
    // (small calls pin little: 1 MiB steps below 16 MiB, 64 MiB steps beyond)
    size_t cap = round_up(bytes ? bytes : 1, bytes < (16u << 20) ? (1u << 20) : (64u << 20));
    { const size_t lim = ring_max_ ? ring_max_ : kRingMax; if (cap > lim && bytes <= lim) cap = lim; }      // (the growth step does not carry the slab over its limit)
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

// ---- the slab as a ring of kPiece slots (moves larger than the slab) ----------------------------------------------------------
// A slot is busy while an H2D copy out of it may still be in flight (uploads return when their copies are ISSUED); slot_ev_[s]
// is recorded behind that copy.  D2H copies into a slot and the workers' memcpys are over when download() returns.
bool Staging::slot_wait(size_t s) {
    if (s < slot_busy_.size() && slot_busy_[s]) {
        if (hipEventSynchronize(slot_ev_[s]) != hipSuccess) return false;
        slot_busy_[s] = 0;
    }
    return true;
}
bool Staging::slot_free_now(size_t s) {
    if (s >= slot_busy_.size() || !slot_busy_[s]) return true;
    const hipError_t e = hipEventQuery(slot_ev_[s]);
    if (e == hipSuccess) { slot_busy_[s] = 0; return true; }
    if (e != hipErrorNotReady) (void)hipGetLastError();
    return false;
}
bool Staging::slot_mark(size_t s, hipStream_t st) {
    if (slot_ev_.size() <= s) { slot_ev_.resize(s + 1, nullptr); slot_busy_.resize(s + 1, 0); }
    if (!slot_ev_[s] && hipEventCreateWithFlags(&slot_ev_[s], hipEventDisableTiming) != hipSuccess) { slot_ev_[s] = nullptr; return false; }
    if (hipEventRecord(slot_ev_[s], st) != hipSuccess) return false;
    slot_busy_[s] = 1;
    return true;
}
bool Staging::drain_slots() {
    for (size_t s = 0; s < slot_busy_.size(); ++s) if (!slot_wait(s)) return false;
    return true;
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
    // DIRECT: the whole address space [0, slab_bytes) fits the slab — a byte's place in the slab is its offset (the callers keep
    // the moves of one call apart by their ranges).  RING: it does not — every task (<= kPiece of the address space, one DMA)
    // borrows the next slot of the ring, and waits for the copy that last read that slot.
    const bool ring = slab_bytes > pin_cap_;
    const size_t nslots = pin_cap_ / kPiece;
    if (ring && nslots < 2) return hipErrorOutOfMemory;
    if (!ring && !drain_slots()) return hipErrorUnknown;
    size_t span_a = SIZE_MAX, span_b = 0;
    for (const Seg& s : segs) if (s.len) { if (s.slab_off < span_a) span_a = s.slab_off; if (s.slab_off + s.len > span_b) span_b = s.slab_off + s.len; }
    // (one copy over the whole span only while the span is mostly payload: sparse segments — small payloads in large slots —
    //  take the piecewise path, which moves what the segments cover)
    if (total <= kInline && span_b - span_a <= 2 * total) {
        size_t slot = 0;
        if (ring) { slot = next_slot_; next_slot_ = (next_slot_ + 1) % nslots; if (!slot_wait(slot)) return hipErrorUnknown; }
        uint8_t* const base = ring ? pin + slot * kPiece - span_a : pin;      // (a byte of the address space at offset o sits at base + o)
        for (const Seg& s : segs) if (s.len) memcpy(base + s.slab_off, s.host, s.len);
        TRY(hipMemcpyAsync(d_base + span_a, base + span_a, span_b - span_a, hipMemcpyHostToDevice, c1));
        ++counters.h2d_copies; counters.h2d_bytes += span_b - span_a;
        if (ring && !slot_mark(slot, c1)) return hipErrorUnknown;
    } else {
        std::vector<Piece> pieces; std::vector<Task> tasks;
        plan(segs, pieces, tasks);
        Pool* p = pool();
        const size_t n = tasks.size();
        std::vector<std::atomic<uint8_t>> done(n);
        for (auto& d : done) d.store(0, std::memory_order_relaxed);
        std::vector<size_t> slot_of(ring ? n : 0);
        auto submit = [&](size_t k) {
            uint8_t* const base = ring ? pin + slot_of[k] * kPiece - tasks[k].a : pin;
            p->submit([&, k, base] {
                const Task& t = tasks[k];
                for (size_t i = t.first; i < t.first + t.count; ++i) memcpy(base + pieces[i].slab_off, pieces[i].host, pieces[i].len);
                done[k].store(1, std::memory_order_release);
            });
        };
        hipError_t err = hipSuccess;
        size_t sub = 0, iss = 0;                       // tasks handed to the workers / tasks whose DMA is issued
        if (!ring) for (; sub < n; ++sub) submit(sub);
        while (iss < n) {
            if (ring) {
                // hand out tasks while the next slot of the ring is free (at most nslots tasks between memcpy and the end of their DMA)
                while (sub < n && sub - iss < nslots && slot_free_now(next_slot_)) { slot_of[sub] = next_slot_; next_slot_ = (next_slot_ + 1) % nslots; submit(sub); ++sub; }
                if (sub == iss) {                      // nothing in the workers' hands: the slot's last copy has to finish first
                    if (!slot_wait(next_slot_)) { err = hipErrorUnknown; break; }
                    continue;
                }
            }
            p->wait_until([&] { return done[iss].load(std::memory_order_acquire) != 0; });
            hipStream_t c = (iss & 1) ? c2 : c1;
            uint8_t* const base = ring ? pin + slot_of[iss] * kPiece - tasks[iss].a : pin;
            if (err == hipSuccess) err = hipMemcpyAsync(d_base + tasks[iss].a, base + tasks[iss].a, tasks[iss].b - tasks[iss].a, hipMemcpyHostToDevice, c);
            ++counters.h2d_copies; counters.h2d_bytes += tasks[iss].b - tasks[iss].a;
            if (ring && err == hipSuccess && !slot_mark(slot_of[iss], c)) err = hipErrorUnknown;
            ++iss;
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
    const bool ring = slab_bytes > pin_cap_;         // (see upload)
    const size_t nslots = pin_cap_ / kPiece;
    if (ring && nslots < 2) return hipErrorOutOfMemory;
    if (!ring && !drain_slots()) return hipErrorUnknown;
    size_t span_a = SIZE_MAX, span_b = 0;
    for (const Seg& s : segs) if (s.len) { if (s.slab_off < span_a) span_a = s.slab_off; if (s.slab_off + s.len > span_b) span_b = s.slab_off + s.len; }
    if (total <= kInline && span_b - span_a <= 2 * total) {
        size_t slot = 0;
        if (ring) { slot = next_slot_; next_slot_ = (next_slot_ + 1) % nslots; if (!slot_wait(slot)) return hipErrorUnknown; }
        uint8_t* const base = ring ? pin + slot * kPiece - span_a : pin;
        TRY(hipMemcpyAsync(base + span_a, d_base + span_a, span_b - span_a, hipMemcpyDeviceToHost, c1));
        ++counters.d2h_copies; counters.d2h_bytes += span_b - span_a;
        TRY(hipStreamSynchronize(c1));
        for (const Seg& s : segs) if (s.len) memcpy(s.host, base + s.slab_off, s.len);
        return hipSuccess;
    }
    std::vector<Piece> pieces; std::vector<Task> tasks;
    plan(segs, pieces, tasks);
    Pool* p = pool();
    const size_t n = tasks.size();
    std::vector<std::atomic<uint8_t>> copied(n);     // the workers have taken task k's bytes out of its slot
    for (auto& d : copied) d.store(0, std::memory_order_relaxed);
    std::vector<size_t> slot_of(ring ? n : 0);
    hipError_t err = hipSuccess;
    size_t iss = 0, hand = 0;                         // tasks whose DMA is issued / tasks handed to the workers
    while (hand < n && err == hipSuccess) {
        // issue DMAs while slots are free: direct, every task has its own place; ring, task k takes the slot task k - nslots has left
        while (iss < n && err == hipSuccess) {
            if (ring) {
                if (iss >= nslots && !copied[iss - nslots].load(std::memory_order_acquire)) break;
                const size_t slot = iss < nslots ? (next_slot_ + iss) % nslots : slot_of[iss - nslots];
                if (iss < nslots && !slot_wait(slot)) { err = hipErrorUnknown; break; }      // (an upload of an earlier call may still read it)
                slot_of[iss] = slot;
            }
            hipEvent_t e = event(2 + iss);
            if (!e) { err = hipErrorUnknown; break; }
            hipStream_t c = (iss & 1) ? c2 : c1;
            uint8_t* const base = ring ? pin + slot_of[iss] * kPiece - tasks[iss].a : pin;
            err = hipMemcpyAsync(base + tasks[iss].a, d_base + tasks[iss].a, tasks[iss].b - tasks[iss].a, hipMemcpyDeviceToHost, c);
            ++counters.d2h_copies; counters.d2h_bytes += tasks[iss].b - tasks[iss].a;
            if (err == hipSuccess) err = hipEventRecord(e, c);
            ++iss;
        }
        if (err != hipSuccess) break;
        if (hand < iss) {
            err = hipEventSynchronize(events_[2 + hand]);
            if (err != hipSuccess) break;
            const size_t k = hand;
            uint8_t* const base = ring ? pin + slot_of[k] * kPiece - tasks[k].a : pin;
            p->submit([&, k, base] {
                const Task& t = tasks[k];
                for (size_t i = t.first; i < t.first + t.count; ++i) memcpy(pieces[i].host, base + pieces[i].slab_off, pieces[i].len);
                copied[k].store(1, std::memory_order_release);
            });
            ++hand;
        } else {
            p->wait_until([&] { return copied[iss - nslots].load(std::memory_order_acquire) != 0; });     // every issued task is with the workers: wait for a slot
        }
    }
    p->wait();
    if (ring) next_slot_ = (next_slot_ + (n < nslots ? n : nslots)) % nslots;
    return err;
}

}  // namespace lzf_host
