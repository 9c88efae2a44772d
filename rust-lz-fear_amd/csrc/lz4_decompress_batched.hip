// lz4_decompress_batched.hip — batched raw::decompress_raw for gfx950, second generation.
//
// Same contract as lz4_decompress.hip (src/raw/decompress.rs:58-138, one wavefront per block),
// but the per-sequence memory round trips of the first kernel are gone:
//
//   * the wave parses up to 64 sequences ahead (wave-uniform scalar walk over a 256-byte
//     register window of the compressed input) and gives sequence j to lane j;
//   * the most recent RING bytes of output live in an LDS ring (ring index == output address
//     mod RING, so 16-byte chunks of the ring line up with 16-byte chunks of HBM).  All of a
//     batch's output is assembled in the ring: literals (lane-parallel, register staged),
//     "far" matches (source older than the ring's intact history: read back from HBM with two
//     16-byte loads per lane) and "near" matches (ring -> ring);
//   * near matches are resolved in rounds against a high-water mark H = start of the first
//     unresolved match: a lane may copy once its whole source lies below H.  Overlapping
//     matches use the period-`offset` form dst[t] = src[t mod offset], so all their reads
//     precede the match as well.  LDS executes a wave's accesses in order, so no barrier or
//     wait separates the rounds;
//   * the finished batch is flushed ring -> HBM with aligned 16-byte stores (coalesced
//     write stream, every output byte written once).
//   * a sequence too large for a batch (> RING/4 bytes) takes a solo path: cooperative
//     HBM -> HBM copies as in the first kernel, then the ring is re-filled from HBM.
//
// Error precedence is the reference's: within a sequence literal EOF / LSIC EOF
// (UnexpectedEnd), MemoryLimitExceeded, ZeroDeduplicationOffset, InvalidDeduplicationOffset
// (decompress.rs:63-75,82-89); across sequences the first one in stream order wins.
#include "lzf_device.h"

namespace lzf {

namespace {

constexpr uint32_t kMaxPosB = 0x7FFFFF00u;
constexpr uint32_t kShort = 32;          // bytes a lane moves by itself; longer runs are cooperative

struct InWindowB {
    const uint8_t* in;
    uint32_t len;
    uint32_t base;
    uint32_t w, wn;
    __device__ __forceinline__ uint32_t fetch(uint32_t b, uint32_t lane) const {
        const uint32_t a = b + lane * 4u;
        uint32_t v = 0;
        if (a + 4u <= len) v = ld4(in + a);
        else if (a < len) { for (uint32_t i = 0; a + i < len; ++i) v |= (uint32_t)in[a + i] << (8u * i); }
        return v;
    }
    __device__ __forceinline__ uint32_t byte(uint32_t p, uint32_t lane) {
        const uint32_t b = p & ~255u;
        if (b != base) {
            if (b == base + 256u) w = wn; else w = fetch(b, lane);
            base = b;
            if (b + 256u < len) wn = fetch(b + 256u, lane);
        }
        const uint32_t d = __builtin_amdgcn_readlane(w, (p >> 2) & 63u);
        return (d >> ((p & 3u) * 8u)) & 0xFFu;
    }
};

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { const uint32_t o = __shfl_xor(v, m); v = o > v ? o : v; }
    return v;
}

// 4 input bytes at pos (pos < len), never reading at or beyond len
__device__ __forceinline__ uint32_t safe_ld4(const uint8_t* in, uint32_t len, uint32_t pos) {
    if (pos + 4u <= len) return ld4(in + pos);
    uint32_t v = 0;
    for (uint32_t i = 0; pos + i < len; ++i) v |= (uint32_t)in[pos + i] << (8u * i);
    return v;
}

}  // namespace

template <int RING>
__global__ __launch_bounds__(64) void lzf_decompress_batched_kernel(
    const lzf_decompress_job* __restrict__ jobs, lzf_job_result* __restrict__ results, uint32_t n_jobs) {
    constexpr uint32_t kMask = RING - 1;
    constexpr uint32_t kSpanMax = RING / 4;            // output bytes one batch may produce
    constexpr uint32_t kNearHist = RING - kSpanMax;    // history before the batch that stays intact in the ring
    __shared__ __attribute__((aligned(16))) uint8_t ring[RING];

    const uint32_t jid = blockIdx.x;
    if (jid >= n_jobs) return;
    const uint32_t lane = threadIdx.x;
    const lzf_decompress_job job = jobs[jid];
    const long long t_start = clock64();

    int status = LZF_OK;
    uint32_t o = 0;
#ifdef LZF_PHASE_TIMING
    long long g_tph[6] = {0, 0, 0, 0, 0, 0}; uint32_t g_pk_hi = 0;
#endif
    if (job.input_len >= kMaxPosB || job.out_existing_len >= kMaxPosB || job.prefix_len >= kMaxPosB) {
        status = LZF_CONTRACT;
    } else {
        const uint8_t* __restrict__ in = job.input;
        const uint8_t* __restrict__ prefix = job.prefix;
        uint8_t* out = job.out;
        const uint32_t len = (uint32_t)job.input_len;
        const uint32_t plen = (uint32_t)job.prefix_len;
        const uint32_t cap = job.out_cap > kMaxPosB ? kMaxPosB : (uint32_t)job.out_cap;
        const uint64_t limit = job.output_limit;
        const uint32_t rb = (uint32_t)(reinterpret_cast<uintptr_t>(out) & 15u);   // ring bias
#define RIDX(x) (((x) + rb) & kMask)

        // ring <- out[a, b)   (b - a <= RING; caller made out[a,b) visible)
        auto ring_fill = [&](uint32_t a, uint32_t b) {
            uint32_t nh = (16u - ((a + rb) & 15u)) & 15u; if (nh > b - a) nh = b - a;
            if (lane < nh) ring[RIDX(a + lane)] = out[a + lane];
            a += nh;
            const uint32_t nchunks = (b - a) >> 4;
            for (uint32_t c = lane; c < nchunks; c += kWave)
                *reinterpret_cast<u32x4*>(&ring[RIDX(a + 16u * c)]) = *reinterpret_cast<const u32x4*>(out + a + 16u * c);
            a += nchunks << 4;
            if (lane < b - a) ring[RIDX(a + lane)] = out[a + lane];
        };
        // out[a, b) <- ring
        auto ring_flush = [&](uint32_t a, uint32_t b) {
            uint32_t nh = (16u - ((a + rb) & 15u)) & 15u; if (nh > b - a) nh = b - a;
            if (lane < nh) out[a + lane] = ring[RIDX(a + lane)];
            a += nh;
            const uint32_t nchunks = (b - a) >> 4;
            for (uint32_t c = lane; c < nchunks; c += kWave)
                *reinterpret_cast<u32x4*>(out + a + 16u * c) = *reinterpret_cast<const u32x4*>(&ring[RIDX(a + 16u * c)]);
            a += nchunks << 4;
            if (lane < b - a) out[a + lane] = ring[RIDX(a + lane)];
        };

        o = (uint32_t)job.out_existing_len;
        uint32_t safe = o;   // out[0, safe) is visible to this wave's global loads
        if (o > 0) ring_fill(o > (uint32_t)RING ? o - RING : 0u, o);   // Vec content on entry = history
        uint32_t p = 0;
        InWindowB win{in, len, 0xFFFFFFFFu, 0u, 0u};

#ifdef LZF_PHASE_TIMING
        long long tph[6] = {0, 0, 0, 0, 0, 0}; long long tq = clock64();
#define PHASE(i) do { const long long tn = clock64(); tph[i] += tn - tq; tq = tn; } while (0)
#else
#define PHASE(i) do { } while (0)
#endif
        while (p < len && status == LZF_OK) {
            // =====================================================================
            // A. parse up to 64 sequences (decompress.rs:61-74), wave-uniform
            // =====================================================================
            const uint32_t ob0 = o;
            uint32_t nseq = 0, o_run = o, maxL = 0;
            uint32_t v_src = 0, v_L = 0, v_M = 0, v_lo = 0, v_fl = 0;   // lane j = sequence j
            int perr = LZF_OK;
            bool solo = false;
            uint32_t s_src = 0, s_L = 0, s_M = 0; bool s_has = false; uint32_t s_offpos = 0;

            while (nseq < kWave && p < len) {
                const uint32_t tp = p;
                const uint32_t token = win.byte(p, lane); ++p;
                uint32_t L = token >> 4;
                if (L == 15u) {
                    bool eof = false;
                    for (;;) {
                        if (p >= len) { eof = true; break; }
                        const uint32_t b = win.byte(p, lane); ++p;
                        L += b; if (L > kMaxPosB) L = kMaxPosB;
                        if (b != 255u) break;
                    }
                    if (eof) { perr = LZF_UNEXPECTED_END; break; }
                }
                if (len - p < L) { perr = LZF_UNEXPECTED_END; break; }            // :67
                const uint32_t src = p;
                uint32_t q = p + L, M = 0; bool has = true;
                if (len - q < 2u) { has = false; q = len; }                      // :70 no match: stream ends
                else {
                    q += 2u;
                    M = token & 15u;
                    if (M == 15u) {
                        bool eof = false;
                        for (;;) {
                            if (q >= len) { eof = true; break; }
                            const uint32_t b = win.byte(q, lane); ++q;
                            M += b; if (M > kMaxPosB) M = kMaxPosB;
                            if (b != 255u) break;
                        }
                        if (eof) { perr = LZF_UNEXPECTED_END; break; }
                    }
                    M += 4u;
                }
                if (cap - o_run < L) { perr = LZF_OUT_CAPACITY; break; }
                if (has && (uint64_t)o_run + L + M > limit) { perr = LZF_MEMORY_LIMIT_EXCEEDED; break; }   // :72-74
                // too large for a batch?
                if ((uint64_t)(o_run - ob0) + L + M > kSpanMax) {
                    if (nseq == 0) { solo = true; s_src = src; s_L = L; s_M = M; s_has = has; s_offpos = src + L; p = q; }
                    else p = tp;          // close the batch before this sequence
                    break;
                }
                uint32_t fl = has ? 1u : 0u;
                if (has && cap - (o_run + L) < M) fl |= 2u;                        // our buffer, lowest precedence
                if (lane == nseq) { v_src = src; v_L = L; v_M = M; v_lo = o_run; v_fl = fl; }   // v_writelane
                if (L > maxL) maxL = L;
                o_run += L + M;
                ++nseq;
                p = q;
                if (fl & 2u) break;       // nothing after a capacity failure matters
            }

            PHASE(0);
            // =====================================================================
            // B. the batch: lane j < nseq owns sequence j
            // =====================================================================
            if (nseq > 0) {
                const bool act = lane < nseq;
                const uint32_t L = act ? v_L : 0u;
                const uint32_t M = act ? v_M : 0u;
                const uint32_t lo = v_lo, src = v_src;
                const uint32_t mo = lo + L;
                const bool has = act && (v_fl & 1u);
                uint32_t off = 0;
                if (has) off = (uint32_t)in[src + L] | ((uint32_t)in[src + L + 1u] << 8);
                // ---- errors, first sequence in stream order wins
                int code = LZF_OK;
                if (has) {
                    if (off == 0u) code = LZF_ZERO_DEDUP_OFFSET;                               // :83
                    else if (off > mo && off - mo > plen) code = LZF_INVALID_DEDUP_OFFSET;    // :84-89
                    else if (v_fl & 2u) code = LZF_OUT_CAPACITY;
                }
                const uint32_t e = first_lane(__ballot(code != LZF_OK));
                if (e < 64u) { status = __builtin_amdgcn_readlane(code, e); break; }
                if (perr != LZF_OK && !solo) {
                    // the failing sequence comes after every lane of this batch; output content is
                    // unspecified on error, so stop here
                    status = perr; break;
                }

                PHASE(1);
                // ---- literals -> ring (decompress.rs:65-67)
                if (maxL > 0u) {
                    const uint32_t Lc = L < kShort ? L : kShort;
                    const uint32_t maxLc = maxL < kShort ? maxL : kShort;
                    uint32_t d[8];
#pragma unroll
                    for (uint32_t k = 0; k < 8u; ++k) {
                        d[k] = 0u;
                        if (4u * k < maxLc) { if (4u * k < Lc) d[k] = safe_ld4(in, len, src + 4u * k); }
                    }
#pragma unroll
                    for (uint32_t t = 0; t < kShort; ++t) {
                        if (t < maxLc) { if (t < Lc) ring[RIDX(lo + t)] = (uint8_t)(d[t >> 2] >> ((t & 3u) * 8u)); }
                    }
                    if (maxL > kShort) {
                        for (unsigned long long m = __ballot(L > kShort); m; m &= m - 1ull) {
                            const uint32_t j = (uint32_t)__builtin_ctzll(m);
                            const uint32_t jl = __builtin_amdgcn_readlane(L, j);
                            const uint32_t js = __builtin_amdgcn_readlane(src, j);
                            const uint32_t jo = __builtin_amdgcn_readlane(lo, j);
                            for (uint32_t i = kShort + lane; i < jl; i += 4u * kWave) {
                                const uint32_t i1 = i + kWave, i2 = i + 2u * kWave, i3 = i + 3u * kWave;
                                const uint8_t b0 = in[js + i];
                                const uint8_t b1 = i1 < jl ? in[js + i1] : (uint8_t)0;
                                const uint8_t b2 = i2 < jl ? in[js + i2] : (uint8_t)0;
                                const uint8_t b3 = i3 < jl ? in[js + i3] : (uint8_t)0;
                                ring[RIDX(jo + i)] = b0;
                                if (i1 < jl) ring[RIDX(jo + i1)] = b1;
                                if (i2 < jl) ring[RIDX(jo + i2)] = b2;
                                if (i3 < jl) ring[RIDX(jo + i3)] = b3;
                            }
                        }
                    }
                }

                PHASE(2);
                // ---- matches (copy_overlapping, decompress.rs:80-138)
                const uint32_t near_lo = ob0 > kNearHist ? ob0 - kNearHist : 0u;
                const uint32_t span = M < off ? M : off;                  // distinct source bytes
                const bool from_prefix = has && off > mo;
                const uint32_t s0 = mo - off;                             // valid when !from_prefix
                const bool is_near = has && !from_prefix && s0 >= near_lo;
                const bool is_far = has && !from_prefix && s0 + span <= near_lo;
                const bool is_slow = has && !is_near && !is_far;          // prefix or straddling
                // HBM visibility of what far / slow lanes read back
                {
                    uint32_t need = 0;
                    if (is_far) need = s0 + (M > kShort ? M : kShort);
                    if (is_slow && !from_prefix) need = near_lo;
                    if (is_slow && from_prefix && M > off - mo) need = near_lo;
                    if (need > ob0) need = ob0;
                    if (wave_max_u32(need) > safe) { wave_store_fence(); safe = ob0; }
                }
                // far, short: two 16-byte loads per lane, bytes into the ring
                if (__ballot(is_far && M <= kShort)) {
                    u32x4 f0 = {0, 0, 0, 0}, f1 = {0, 0, 0, 0};
                    const bool go = is_far && M <= kShort;
                    if (go) { f0 = ld16(out + s0); if (M > 16u) f1 = ld16(out + s0 + 16u); }
                    const uint32_t fm = go ? M : 0u;
                    const uint32_t maxfm = wave_max_u32(fm);
#pragma unroll
                    for (uint32_t t = 0; t < kShort; ++t) {
                        if (t < maxfm) {
                            const uint32_t dw = t < 16u ? f0[(t >> 2) & 3u] : f1[(t >> 2) & 3u];
                            if (t < fm) ring[RIDX(mo + t)] = (uint8_t)(dw >> ((t & 3u) * 8u));
                        }
                    }
                }
                // far, long: cooperative HBM -> ring (never overlapping: offset > ring history > length)
                for (unsigned long long m = __ballot(is_far && M > kShort); m; m &= m - 1ull) {
                    const uint32_t j = (uint32_t)__builtin_ctzll(m);
                    const uint32_t jm = __builtin_amdgcn_readlane(M, j);
                    const uint32_t js = __builtin_amdgcn_readlane(s0, j);
                    const uint32_t jo = __builtin_amdgcn_readlane(mo, j);
                    for (uint32_t i = lane; i < jm; i += 4u * kWave) {
                        const uint32_t i1 = i + kWave, i2 = i + 2u * kWave, i3 = i + 3u * kWave;
                        const uint8_t b0 = out[js + i];
                        const uint8_t b1 = i1 < jm ? out[js + i1] : (uint8_t)0;
                        const uint8_t b2 = i2 < jm ? out[js + i2] : (uint8_t)0;
                        const uint8_t b3 = i3 < jm ? out[js + i3] : (uint8_t)0;
                        ring[RIDX(jo + i)] = b0;
                        if (i1 < jm) ring[RIDX(jo + i1)] = b1;
                        if (i2 < jm) ring[RIDX(jo + i2)] = b2;
                        if (i3 < jm) ring[RIDX(jo + i3)] = b3;
                    }
                }
                PHASE(3);
                // near + slow: rounds against the high-water mark
                unsigned long long unresolved = __ballot(is_near || is_slow);
                const unsigned long long slow_mask = __ballot(is_slow);
                // round 1 (lane-parallel): every near lane whose source lies below the first unresolved
                // match start H — on typical data that is almost all of them
                if (unresolved) {
                    const uint32_t f = (uint32_t)__builtin_ctzll(unresolved);
                    const uint32_t H = __builtin_amdgcn_readlane(mo, f);      // everything below H is final
                    const bool ready = is_near && (s0 + span <= H) && !((slow_mask >> f) & 1ull);
                    const unsigned long long rmask = __ballot(ready);
                    const uint32_t cnt = (ready && M <= kShort) ? M : 0u;
                    const uint32_t maxcnt = wave_max_u32(cnt);
                    uint32_t r = 0;
                    for (uint32_t t0 = 0; t0 < maxcnt; t0 += 8u) {
                        uint8_t b[8];
#pragma unroll
                        for (uint32_t k = 0; k < 8u; ++k) {
                            b[k] = 0;
                            if (t0 + k < cnt) { b[k] = ring[RIDX(s0 + r)]; ++r; if (r == off) r = 0u; }
                        }
#pragma unroll
                        for (uint32_t k = 0; k < 8u; ++k) {
                            if (t0 + k < cnt) ring[RIDX(mo + t0 + k)] = b[k];
                        }
                    }
                    // long ready lanes go through the in-order loop below (their sources are final)
                    unresolved &= ~(rmask & __ballot(M <= kShort));
                }
                // the rest strictly in stream order, one sequence at a time, all lanes on it
                while (unresolved) {
                    const uint32_t f = (uint32_t)__builtin_ctzll(unresolved);
                    unresolved &= unresolved - 1ull;
                    if ((slow_mask >> f) & 1ull) {
                        // prefix / straddling source: one lane, byte-serial, three sources
                        if (lane == f) {
                            for (uint32_t t = 0; t < M; ++t) {
                                uint8_t v;
                                if (off > mo + t) v = prefix[plen - (off - mo) + t];                // :91-93
                                else {
                                    const uint32_t s = mo + t - off;
                                    v = s < near_lo ? out[s] : ring[RIDX(s)];
                                }
                                ring[RIDX(mo + t)] = v;
                            }
                        }
                        continue;
                    }
                    const uint32_t jm = __builtin_amdgcn_readlane(M, f);
                    const uint32_t js = __builtin_amdgcn_readlane(s0, f);
                    const uint32_t jo = __builtin_amdgcn_readlane(mo, f);
                    const uint32_t joff = __builtin_amdgcn_readlane(off, f);
                    if (jm <= joff) {                                   // non-overlapping: 4 bytes in flight per lane
                        for (uint32_t i = lane; i < jm; i += 4u * kWave) {
                            const uint32_t i1 = i + kWave, i2 = i + 2u * kWave, i3 = i + 3u * kWave;
                            const uint8_t b0 = ring[RIDX(js + i)];
                            const uint8_t b1 = i1 < jm ? ring[RIDX(js + i1)] : (uint8_t)0;
                            const uint8_t b2 = i2 < jm ? ring[RIDX(js + i2)] : (uint8_t)0;
                            const uint8_t b3 = i3 < jm ? ring[RIDX(js + i3)] : (uint8_t)0;
                            ring[RIDX(jo + i)] = b0;
                            if (i1 < jm) ring[RIDX(jo + i1)] = b1;
                            if (i2 < jm) ring[RIDX(jo + i2)] = b2;
                            if (i3 < jm) ring[RIDX(jo + i3)] = b3;
                        }
                    } else {                                            // overlapping: period-`offset` addressing
                        uint32_t rr = lane % joff;
                        const uint32_t adv = kWave % joff;
                        for (uint32_t i = lane; i < jm; i += kWave) {
                            ring[RIDX(jo + i)] = ring[RIDX(js + rr)];
                            rr += adv; if (rr >= joff) rr -= joff;
                        }
                    }
                }
                PHASE(4);
                // ---- flush the batch ring -> HBM
                o = o_run;
                ring_flush(ob0, o);
                PHASE(5);
            }
            if (perr != LZF_OK && !solo) { status = perr; break; }

            // =====================================================================
            // C. solo sequence (larger than a batch): HBM -> HBM, then re-fill the ring
            // =====================================================================
            if (solo) {
                const uint32_t o_before = o;
                wave_copy(out + o, in + s_src, s_L, lane);                           // literals
                o += s_L;
                if (s_has) {
                    const uint32_t offset = win.byte(s_offpos, lane) | (win.byte(s_offpos + 1u, lane) << 8);
                    uint32_t mlen = s_M;
                    if (offset == 0u) { status = LZF_ZERO_DEDUP_OFFSET; break; }
                    bool done = false;
                    if (offset > o) {
                        const uint32_t need = offset - o;
                        if (need > plen) { status = LZF_INVALID_DEDUP_OFFSET; break; }
                        const uint32_t n = need < mlen ? need : mlen;
                        if (cap - o < n) { status = LZF_OUT_CAPACITY; break; }
                        wave_copy(out + o, prefix + (plen - need), n, lane);
                        o += n; mlen -= n;
                        done = mlen == 0u;
                    }
                    if (!done) {
                        if (cap - o < mlen) { status = LZF_OUT_CAPACITY; break; }
                        const uint32_t src0 = o - offset;
                        const uint32_t span = mlen < offset ? mlen : offset;
                        if (src0 + span > safe) { wave_store_fence(); safe = o; }
                        const uint8_t* hist = out + src0;
                        uint8_t* dst = out + o;
                        if (mlen <= offset) {
                            wave_copy(dst, hist, mlen, lane);
                        } else if (offset == 1u) {
                            const uint32_t b = hist[0];
                            const uint32_t b4 = b * 0x01010101u;
                            const u32x4 v = {b4, b4, b4, b4};
                            const uint32_t bulk = mlen & ~15u;
                            for (uint32_t i = lane * 16u; i < bulk; i += kWave * 16u) st16(dst + i, v);
                            if (lane < mlen - bulk) dst[bulk + lane] = (uint8_t)b;
                        } else {
                            uint32_t r = lane % offset;
                            const uint32_t adv = kWave % offset;
                            for (uint32_t i = lane; i < mlen; i += kWave) {
                                dst[i] = hist[r];
                                r += adv; if (r >= offset) r -= offset;
                            }
                        }
                        o += mlen;
                    }
                }
                // ring <- the tail of what was just written
                wave_store_fence(); safe = o;
                const uint32_t a = (o - o_before > (uint32_t)RING) ? o - RING : o_before;
                ring_fill(a, o);
                if (perr != LZF_OK) { status = perr; break; }
            }
        }
#ifdef LZF_PHASE_TIMING
        for (int i = 0; i < 6; ++i) g_tph[i] = tph[i];
#endif
#undef RIDX
    }
    if (lane == 0) {
        results[jid].out_len = o;
#ifdef LZF_PHASE_TIMING
        // debug build only: six phase totals, 10 bits each in units of 2^20 cycles, above bit 32 / in `reserved`
        {
            unsigned long long pk = 0;
            for (int i = 0; i < 6; ++i) { unsigned long long u = (unsigned long long)(g_tph[i] >> 20); if (u > 1023) u = 1023; pk |= u << (10 * i); }
            results[jid].out_len = (unsigned long long)o | ((pk & 0xFFFFFFFFull) << 32);
            g_pk_hi = (uint32_t)(pk >> 32);
        }
#endif
        results[jid].status = status;
#ifdef LZF_PHASE_TIMING
        results[jid].reserved = g_pk_hi;
#else
        results[jid].reserved = (uint32_t)((clock64() - t_start) >> 10);   // diagnostic: shader kilo-cycles spent on this job
#endif
    }
}

template __global__ void lzf_decompress_batched_kernel<16384>(const lzf_decompress_job*, lzf_job_result*, uint32_t);
template __global__ void lzf_decompress_batched_kernel<8192>(const lzf_decompress_job*, lzf_job_result*, uint32_t);
template __global__ void lzf_decompress_batched_kernel<32768>(const lzf_decompress_job*, lzf_job_result*, uint32_t);

}  // namespace lzf
