// host_staging.h — how host buffers reach HBM and come back (the *_host helpers of capi.hip and the frame layer).
//
// The reference's frame layer streams io::Read -> Vec -> io::Write one block at a time (src/framed/compress.rs:222-263,
// src/framed/decompress.rs:198-279).  Here a call moves all of its blocks at once, so the moves are built for PCIe:
//   * one pinned slab (kept between calls, grown on demand up to 2 GiB) is the only memory the DMA engines touch; a move larger
//     than the slab goes through it as a ring of 4 MiB slots (a slot is reused when the copy that last read it has finished), so
//     the size of a pass is bounded by device memory, not by pinned host memory;
//   * worker threads copy the caller's pageable buffers into / out of the slab in pieces of a few MiB while the calling
//     thread issues one asynchronous H2D / D2H copy per finished piece — the memcpy of piece k + 1 overlaps the DMA of k;
//   * device scratch is kept between calls as well (hipMalloc / hipFree of gigabytes cost more than the kernels).
// Every HIP call is made by the calling thread (its current device is the one used); workers only run memcpy.
// One instance per device; one call at a time per device uses its slab: the entry points take Staging::lock().
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <mutex>
#include <vector>

namespace lzf_host {

// A pageable host range and its place in the slab.
struct Seg { size_t slab_off; uint8_t* host; size_t len; };

class Staging {
public:
    static Staging& get();
    std::mutex& lock() { return mu_; }

    // ---- memory kept between calls (all may fail: false + hip error left for hipGetLastError) ----
    uint8_t* pinned(size_t bytes);                     // the pinned slab, at least `bytes` long
    uint8_t* mailbox(size_t bytes);                    // a second, small pinned buffer for results that come back asynchronously
    void* device(int slot, size_t bytes);              // device scratch slot (0..kSlots-1), at least `bytes` long
    static constexpr int kSlots = 40;
    hipStream_t stream(int i);                         // 0: compute, 1..2: copies in, 3: checksums, 4..5: copies out (non-blocking streams)
    void release();                                    // give everything back (lzf_host_release_scratch)
    void set_threads(unsigned n);                      // worker threads for the next calls (0 = default)
    void set_pinned_limit(size_t bytes);               // the slab stops growing here (0 = default 2 GiB; at least two slots); gives the current slab back
    size_t pinned_capacity() const { return pin_cap_; }

    // ---- moves ----
    // host -> slab -> device: segs[i].host[0..len) lands at d_base + segs[i].slab_off.  Asynchronous on the copy streams
    // (join_copies makes another stream wait for them).  The slab range in use is [0, slab_bytes).
    hipError_t upload(const std::vector<Seg>& segs, size_t slab_bytes, uint8_t* d_base);
    hipError_t join_copies(hipStream_t waiter);        // `waiter` waits for everything issued on the copy streams so far
    // device -> slab -> host: waits for `before` (a stream, may be null) and / or `ready` (an event already recorded, may be
    // null), then the mirror image; returns when every byte is in place.
    hipError_t download(const std::vector<Seg>& segs, size_t slab_bytes, const uint8_t* d_base, hipStream_t before, hipEvent_t ready = nullptr);
    // fn(i) for i in [0, n) on the workers (and the calling thread); returns when all are done
    void parallel_for(size_t n, const std::function<void(size_t)>& fn);

private:
    Staging() = default;
    struct Pool;
    Pool* pool();
    hipEvent_t event(size_t i);
    // the slab as a ring of slots (host_staging.cpp): moves larger than the slab
    bool slot_wait(size_t s);
    bool slot_free_now(size_t s);
    bool slot_mark(size_t s, hipStream_t st);
    bool drain_slots();
    std::vector<hipEvent_t> slot_ev_; std::vector<uint8_t> slot_busy_; size_t next_slot_ = 0;
    size_t ring_max_ = 0;                              // 0: kRingMax
    std::mutex mu_;
    uint8_t* pin_ = nullptr; size_t pin_cap_ = 0;
    uint8_t* mail_ = nullptr; size_t mail_cap_ = 0;
    void* dev_[kSlots] = {}; size_t dev_cap_[kSlots] = {};
    hipStream_t streams_[6] = {};
    std::vector<hipEvent_t> events_;
public:
    struct Counters { uint64_t h2d_copies, d2h_copies, h2d_bytes, d2h_bytes; } counters = {0, 0, 0, 0};
private:
    Pool* pool_ = nullptr;
    static constexpr unsigned kNoThreads = 0xFFFFFFFFu;      // set_threads(LZF_HOST_THREADS_NONE): no worker threads
    unsigned want_threads_ = 0;
    int device_ = -1;
};

}  // namespace lzf_host
