/*
 * lzf_oracle.c — CPU restatement of lz-fear 0.2.0 (raw LZ4 block codec + frame layer).
 * TEST INFRASTRUCTURE ONLY — see lzf_oracle.h for the usage rule and the parity pins.
 * Every function cites the reference lines it follows (paths relative to the lz-fear tree).
 */
#include "lzf_oracle.h"
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * helpers
 * ---------------------------------------------------------------------------------------- */
static inline uint32_t rd32le(const uint8_t* p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static inline uint64_t rd64le(const uint8_t* p) {
    return (uint64_t)rd32le(p) | ((uint64_t)rd32le(p + 4) << 32);
}
static inline void wr32le(uint8_t* p, uint32_t v) {
    p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
}

/* ------------------------------------------------------------------------------------------
 * hashes — src/raw/compress/mod.rs:40-61 (64-bit little-endian target)
 * ---------------------------------------------------------------------------------------- */
/* mod.rs:41-51: read 8 bytes if available, ELSE 0 (quirk B3); ((v<<24)*889523592379)>>52 */
static inline uint32_t hash_for_u32(const uint8_t* p, size_t remaining) {
    uint64_t v = remaining >= 8 ? rd64le(p) : 0;
    return (uint32_t)(((v << 24) * 889523592379ULL) >> (64 - 12));
}
/* mod.rs:58-61: (u32 * 2654435761) >> (32-12-1) -> 13 bits */
static inline uint32_t hash_for_u16(const uint8_t* p) {
    return (rd32le(p) * 2654435761U) >> (32 - 12 - 1);
}

/* mod.rs:63-76 */
size_t lzfo_u32_replace(lzfo_u32_table* t, const uint8_t* input, size_t len, size_t pos, int* contract) {
    uint64_t o = (uint64_t)pos + t->offset;                 /* :65 */
    if (o > 0xFFFFFFFFull) { *contract = 1; return 0; }     /* :67 try_into().expect() */
    uint32_t h = hash_for_u32(input + pos, len - pos);      /* :68 */
    uint64_t old = t->dict[h];
    t->dict[h] = (uint32_t)o;
    return old > t->offset ? (size_t)(old - t->offset) : 0; /* :70 saturating_sub */
}
/* mod.rs:88-101 */
size_t lzfo_u16_replace(lzfo_u16_table* t, const uint8_t* input, size_t len, size_t pos, int* contract) {
    uint64_t o = (uint64_t)pos + t->offset;                 /* :90 */
    if (o > 0xFFFFull || len - pos < 4) { *contract = 1; return 0; } /* :92, read_u32 panic */
    uint32_t h = hash_for_u16(input + pos);                 /* :93 */
    uint64_t old = t->dict[h];
    t->dict[h] = (uint16_t)o;
    return old > t->offset ? (size_t)(old - t->offset) : 0; /* :95 */
}

/* ------------------------------------------------------------------------------------------
 * bounded sink — src/framed/compress.rs:294-314 (NoPartialWrites): a write either fits
 * completely or fails without writing.  An unbounded Vec writer is the cap = SIZE_MAX case.
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint8_t* p; size_t pos, cap; } sink_t;
static inline int sink_write(sink_t* s, const uint8_t* d, size_t n) {
    if (s->cap - s->pos < n) return 0;                      /* :298-301 */
    memcpy(s->p + s->pos, d, n);
    s->pos += n;
    return 1;
}
static inline int sink_u8(sink_t* s, uint8_t b) { return sink_write(s, &b, 1); }

/* mod.rs:243-260 write_lsic_tail */
static int write_lsic_tail(sink_t* s, size_t value) {
    static const uint8_t ff4[4] = {0xFF, 0xFF, 0xFF, 0xFF};
    if (value < 0xF) return 1;                              /* :245 */
    value -= 0xF;
    while (value >= 4 * 0xFF) {                             /* :251-254, one 4-byte write */
        if (!sink_write(s, ff4, 4)) return 0;
        value -= 4 * 0xFF;
    }
    while (value >= 0xFF) {                                 /* :255-258 */
        if (!sink_u8(s, 0xFF)) return 0;
        value -= 0xFF;
    }
    return sink_u8(s, (uint8_t)value);                      /* :259 */
}

/* mod.rs:117-145 count_matching_bytes: common prefix of a (|a| = alen) and b (longer). */
static size_t count_matching_bytes(const uint8_t* a, size_t alen, const uint8_t* b, size_t blen) {
    size_t n = alen < blen ? alen : blen;
    size_t m = 0;
    size_t chunks = (alen / 8 < blen / 8) ? alen / 8 : blen / 8;   /* :129 chunks_exact zip */
    for (size_t c = 0; c < chunks; c++) {
        uint64_t x = rd64le(a + 8 * c) ^ rd64le(b + 8 * c);
        if (x == 0) { m += 8; }
        else { return m + (size_t)(__builtin_ctzll(x) / 8); }      /* :136-137 */
    }
    while (m < n && a[m] == b[m]) m++;                              /* :143 */
    return m;
}

/* ------------------------------------------------------------------------------------------
 * compress2 — src/raw/compress/mod.rs:165-238
 * ---------------------------------------------------------------------------------------- */
int lzfo_compress2(const uint8_t* input, size_t len, size_t cursor, int kind, void* table,
                   uint8_t* out, size_t cap, size_t* out_len) {
    sink_t s = {out, 0, cap};
    int contract = 0;
    size_t limit = kind == LZFO_TABLE_U16 ? 0xFFFFu : 0xFFFFFFFFull;   /* :75, :100 */
    *out_len = 0;
    if (len > limit) return LZFO_CONTRACT;                             /* :167 (cursor >= len: the loop at :171 never runs -> Ok, nothing written) */

#define REPLACE(pos) (kind == LZFO_TABLE_U16                                          \
        ? lzfo_u16_replace((lzfo_u16_table*)table, input, len, (pos), &contract)      \
        : lzfo_u32_replace((lzfo_u32_table*)table, input, len, (pos), &contract))

    const size_t init_cursor = cursor;                                 /* :169 */
    while (cursor < len) {                                             /* :171 (B4: empty -> nothing) */
        const size_t literal_start = cursor;                           /* :172 */
        size_t step_counter = 1u << 6;                                 /* :174 */
        size_t step = 1;                                               /* :175 */
        size_t dup_offset = 0, extra_bytes = 0;
        for (;;) {
            size_t left = len > cursor ? len - cursor : 0;             /* :178 saturating_sub */
            if (left < 12) {
                size_t literal_len = len - literal_start;              /* :182 */
                uint8_t token = (uint8_t)((literal_len < 0xF ? literal_len : 0xF) << 4);
                if (!sink_u8(&s, token)) goto full;                    /* :186 */
                if (!write_lsic_tail(&s, literal_len)) goto full;      /* :187 */
                if (!sink_write(&s, input + literal_start, literal_len)) goto full;  /* :188 */
                *out_len = s.pos;
                return LZFO_OK;                                        /* :189 */
            }
            size_t candidate = REPLACE(cursor);                        /* :196 */
            if (contract) return LZFO_CONTRACT;
            if (cursor != init_cursor && candidate <= cursor && cursor - candidate <= 0xFFFF) { /* :200-201 */
                size_t m = count_matching_bytes(input + cursor, (len - 5) - cursor,  /* :195 */
                                                input + candidate, len - candidate); /* :203 */
                if (m >= 4) {                                          /* :206 */
                    dup_offset = cursor - candidate;                   /* :208 */
                    size_t max_backtrack = cursor - literal_start;     /* :211 */
                    size_t bt = 0;                                     /* :212 */
                    while (bt < max_backtrack && bt < candidate &&
                           input[cursor - 1 - bt] == input[candidate - 1 - bt]) bt++;
                    extra_bytes = m - 4 + bt;                          /* :206,:214 */
                    cursor += m;                                       /* :215 */
                    (void)REPLACE(cursor - 2);                         /* :218 (B1, B3) */
                    if (contract) return LZFO_CONTRACT;
                    break;                                             /* :220 */
                }
            }
            cursor += step;                                            /* :225 */
            step = step_counter >> 6;                                  /* :226 */
            if (literal_start + 1 != cursor) step_counter += 1;        /* :229-231 */
        }
        /* :235-236 + write_group :150-163 */
        size_t literal_end = cursor - extra_bytes - 4;
        size_t literal_len = literal_end - literal_start;
        uint8_t token = (uint8_t)(((literal_len < 0xF ? literal_len : 0xF) << 4) |
                                  (extra_bytes < 0xF ? extra_bytes : 0xF));
        uint8_t off2[2] = {(uint8_t)dup_offset, (uint8_t)(dup_offset >> 8)};
        if (!sink_u8(&s, token)) goto full;                            /* :158 */
        if (!write_lsic_tail(&s, literal_len)) goto full;              /* :159 */
        if (!sink_write(&s, input + literal_start, literal_len)) goto full; /* :160 */
        if (!sink_write(&s, off2, 2)) goto full;                       /* :161 */
        if (!write_lsic_tail(&s, extra_bytes)) goto full;              /* :162 */
    }
    *out_len = s.pos;
    return LZFO_OK;
full:
    *out_len = s.pos;
    return LZFO_OUTPUT_FULL;
#undef REPLACE
}

/* ------------------------------------------------------------------------------------------
 * decompress_raw — src/raw/decompress.rs:58-138
 * ---------------------------------------------------------------------------------------- */
int lzfo_decompress_raw(const uint8_t* in, size_t len, const uint8_t* prefix, size_t prefix_len,
                        uint8_t* out, size_t* out_len, size_t out_cap, size_t output_limit) {
    size_t pos = 0, o = *out_len;
    int rc = LZFO_OK;
    while (pos < len) {                                    /* :61 while let Ok(token) */
        uint8_t token = in[pos++];
        size_t lit = token >> 4;                           /* :63 read_lsic :30-43 */
        if (lit == 0xF) {
            for (;;) {
                if (pos >= len) { rc = LZFO_UNEXPECTED_END; goto done; }
                uint8_t more = in[pos++];
                lit += more;
                if (more != 0xFF) break;
            }
        }
        if (len - pos < lit) { rc = LZFO_UNEXPECTED_END; goto done; }   /* :67 read_exact */
        if (out_cap - o < lit) { rc = LZFO_OUT_CAPACITY; goto done; }   /* (oracle buffer only) */
        memcpy(out + o, in + pos, lit);                    /* :65-67, literals NOT limit-checked */
        pos += lit; o += lit;

        if (len - pos < 2) { pos = len; continue; }        /* :70 read_u16 Err -> no match; the
                                                              failed Cursor::read_exact leaves the
                                                              cursor at EOF, so the loop ends */
        size_t offset = (size_t)in[pos] | ((size_t)in[pos + 1] << 8);
        pos += 2;
        size_t mlen = token & 0xF;                         /* :71 */
        if (mlen == 0xF) {
            for (;;) {
                if (pos >= len) { rc = LZFO_UNEXPECTED_END; goto done; }
                uint8_t more = in[pos++];
                mlen += more;
                if (more != 0xFF) break;
            }
        }
        mlen += 4;
        if (o + mlen > output_limit) { rc = LZFO_MEMORY_LIMIT_EXCEEDED; goto done; }  /* :72-74 */
        /* copy_overlapping :80-138 — every arm is observationally the byte-serial copy */
        if (offset == 0) { rc = LZFO_ZERO_DEDUP_OFFSET; goto done; }    /* :83 */
        if (offset > o) {                                  /* :84-99 */
            size_t need = offset - o;
            if (need > prefix_len) { rc = LZFO_INVALID_DEDUP_OFFSET; goto done; }  /* :87-89 */
            size_t n = need < mlen ? need : mlen;          /* :90 */
            if (out_cap - o < n) { rc = LZFO_OUT_CAPACITY; goto done; }
            memcpy(out + o, prefix + (prefix_len - need), n);
            o += n; mlen -= n;                             /* :94-98 recurse with empty prefix */
        }
        if (out_cap - o < mlen) { rc = LZFO_OUT_CAPACITY; goto done; }
        for (size_t i = 0; i < mlen; i++) out[o + i] = out[o - offset + i];   /* :128-135 */
        o += mlen;
    }
done:
    *out_len = o;
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * XXH32 (public specification; twox-hash XxHash32, Cargo.toml:17)
 * ---------------------------------------------------------------------------------------- */
#define XP1 2654435761U
#define XP2 2246822519U
#define XP3 3266489917U
#define XP4 668265263U
#define XP5 374761393U
static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static inline uint32_t xround(uint32_t acc, uint32_t in) { return rotl32(acc + in * XP2, 13) * XP1; }

typedef struct { uint32_t v[4]; uint8_t buf[16]; uint32_t buf_len; uint64_t total; uint32_t seed; } xxh32_state;
static void xxh32_init(xxh32_state* st, uint32_t seed) {
    st->v[0] = seed + XP1 + XP2; st->v[1] = seed + XP2; st->v[2] = seed; st->v[3] = seed - XP1;
    st->buf_len = 0; st->total = 0; st->seed = seed;
}
static void xxh32_update(xxh32_state* st, const uint8_t* p, size_t len) {
    st->total += len;
    if (st->buf_len) {
        size_t take = 16 - st->buf_len; if (take > len) take = len;
        memcpy(st->buf + st->buf_len, p, take);
        st->buf_len += (uint32_t)take; p += take; len -= take;
        if (st->buf_len < 16) return;
        for (int i = 0; i < 4; i++) st->v[i] = xround(st->v[i], rd32le(st->buf + 4 * i));
        st->buf_len = 0;
    }
    while (len >= 16) {
        for (int i = 0; i < 4; i++) st->v[i] = xround(st->v[i], rd32le(p + 4 * i));
        p += 16; len -= 16;
    }
    if (len) { memcpy(st->buf, p, len); st->buf_len = (uint32_t)len; }
}
static uint32_t xxh32_digest(const xxh32_state* st) {
    uint32_t h;
    if (st->total >= 16) h = rotl32(st->v[0], 1) + rotl32(st->v[1], 7) + rotl32(st->v[2], 12) + rotl32(st->v[3], 18);
    else h = st->seed + XP5;
    h += (uint32_t)st->total;
    const uint8_t* p = st->buf; uint32_t n = st->buf_len;
    while (n >= 4) { h = rotl32(h + rd32le(p) * XP3, 17) * XP4; p += 4; n -= 4; }
    while (n) { h = rotl32(h + (*p) * XP5, 11) * XP1; p++; n--; }
    h ^= h >> 15; h *= XP2; h ^= h >> 13; h *= XP3; h ^= h >> 16;
    return h;
}
uint32_t lzfo_xxh32(const uint8_t* p, size_t len, uint32_t seed) {
    xxh32_state st; xxh32_init(&st, seed); xxh32_update(&st, p, len); return xxh32_digest(&st);
}

/* ------------------------------------------------------------------------------------------
 * frame layer — src/framed/{mod,header,compress,decompress}.rs
 * ---------------------------------------------------------------------------------------- */
#define F_MAGIC 0x184D2204u          /* framed/mod.rs:16 */
#define F_INCOMPRESSIBLE 0x80000000u /* framed/mod.rs:18 */
#define F_WINDOW 65536u              /* framed/mod.rs:20 */
/* header.rs:8-16 */
#define FL_INDEP 0x20
#define FL_BLOCKSUM 0x10
#define FL_CSIZE 0x08
#define FL_CSUM 0x04
#define FL_DICTID 0x01

void lzfo_settings_default(lzfo_settings* s) {            /* framed/compress.rs:44-55 */
    memset(s, 0, sizeof *s);
    s->independent_blocks = 1; s->block_checksums = 0; s->content_checksum = 1;
    s->block_size = 4u * 1024 * 1024;
}

/* header.rs:53-62 BlockDescriptor::new.  rc: LZFO_OK / INVALID_BLOCK_SIZE / PANIC */
static int bd_new(uint64_t block_maxsize, uint8_t* bd) {
    unsigned tz = block_maxsize ? (unsigned)__builtin_ctzll(block_maxsize) : 64;
    unsigned maybe = ((tz > 8 ? tz - 8 : 0) / 2) & 0xFF;            /* :54 */
    uint8_t b = (uint8_t)(maybe << 4);                              /* u8 shift, bits fall off */
    if (b & 0x8F) return LZFO_F_PANIC;                              /* :55 parse().unwrap() */
    unsigned size = (b >> 4) & 7;                                   /* :74 */
    if (size < 4 || size > 7) return LZFO_F_INVALID_BLOCK_SIZE;     /* :57-59 */
    if (((uint64_t)1 << (size * 2 + 8)) != block_maxsize) return LZFO_F_INVALID_BLOCK_SIZE;
    *bd = b;
    return LZFO_OK;
}

size_t lzfo_frame_compress_bound(const lzfo_settings* s, size_t in_len) {
    size_t bs = s->block_size ? (size_t)s->block_size : 1;
    size_t blocks = in_len / bs + 1;
    return 19 + in_len + blocks * 8 + 8;
}

/* framed/compress.rs:160-282 compress_internal */
int lzfo_frame_compress(const lzfo_settings* st, const uint8_t* in, size_t in_len,
                        uint8_t* out, size_t out_cap, size_t* out_len) {
    *out_len = 0;
    uint8_t flags = 0;                                              /* :163-179 */
    if (st->independent_blocks) flags |= FL_INDEP;
    if (st->block_checksums) flags |= FL_BLOCKSUM;
    if (st->content_checksum) flags |= FL_CSUM;
    if (st->has_dictionary_id) flags |= FL_DICTID;
    if (st->has_content_size) flags |= FL_CSIZE;
    uint8_t bd;
    int rc = bd_new(st->block_size, &bd);                           /* :183 */
    if (rc != LZFO_OK) return rc;
    if (out_cap < lzfo_frame_compress_bound(st, in_len)) return LZFO_OUT_CAPACITY;

    size_t w = 0;
    wr32le(out + w, F_MAGIC); w += 4;                               /* :186 */
    out[w++] = (uint8_t)((1 << 6) | flags);                         /* :181-182,:187 */
    out[w++] = bd;                                                  /* :188 */
    if (st->has_content_size) {                                     /* :190-192 */
        wr32le(out + w, (uint32_t)st->content_size); wr32le(out + w + 4, (uint32_t)(st->content_size >> 32)); w += 8;
    }
    if (st->has_dictionary_id) { wr32le(out + w, st->dictionary_id); w += 4; }   /* :193-195 */
    out[w] = (uint8_t)(lzfo_xxh32(out + 4, w - 4, 0) >> 8); w++;    /* :197-199 */

    lzfo_u32_table template_table, table;                           /* :202 */
    memset(&template_table, 0, sizeof template_table);
    const uint8_t* dict = st->dictionary; size_t dict_len = dict ? (size_t)st->dictionary_len : 0;
    int contract = 0;
    if (dict && dict_len >= 8) {                                    /* :204-211 windows(8).step_by(3) */
        for (size_t o = 0; o + 8 <= dict_len; o += 3) lzfo_u32_replace(&template_table, dict, dict_len, o, &contract);
    }
    if (contract) return LZFO_CONTRACT;

    const size_t bs = (size_t)st->block_size;
    xxh32_state content; xxh32_init(&content, 0);
    memcpy(&table, &template_table, sizeof table);                  /* :220 */

    /* in_buffer (:217-218): prefix ++ block, at most dict_len + 64 KiB + block_size bytes */
    size_t ib_cap = dict_len + F_WINDOW + bs + 16;
    uint8_t* in_buffer = (uint8_t*)malloc(ib_cap);
    if (!in_buffer) return LZFO_OUT_CAPACITY;
    size_t ib_len = 0;
    if (dict_len) { memcpy(in_buffer, dict, dict_len); ib_len = dict_len; }   /* :218 */

    size_t rp = 0;
    for (;;) {                                                      /* :221 */
        size_t window_offset = ib_len;                              /* :222 */
        size_t read_bytes = in_len - rp < bs ? in_len - rp : bs;    /* :227 */
        if (read_bytes == 0) break;                                 /* :229-231 */
        memcpy(in_buffer + ib_len, in + rp, read_bytes); ib_len += read_bytes;
        if (st->content_checksum) xxh32_update(&content, in + rp, read_bytes);   /* :233-235 */
        rp += read_bytes;

        size_t clen = 0;
        uint8_t* dst = out + w + 4;
        rc = lzfo_compress2(in_buffer, ib_len, window_offset, LZFO_TABLE_U32, &table,
                            dst, read_bytes /* cap = N, :242 */, &clen);         /* :243 */
        const uint8_t* written; size_t written_len;
        if (rc == LZFO_OK) {                                        /* :244-249 */
            wr32le(out + w, (uint32_t)clen);
            written = dst; written_len = clen;
        } else if (rc == LZFO_OUTPUT_FULL) {                        /* :250-255 */
            wr32le(out + w, (uint32_t)read_bytes | F_INCOMPRESSIBLE);
            memcpy(dst, in_buffer + window_offset, read_bytes);
            written = dst; written_len = read_bytes;
        } else { free(in_buffer); return rc; }
        w += 4 + written_len;                                       /* :258 */
        if (flags & FL_BLOCKSUM) { wr32le(out + w, lzfo_xxh32(written, written_len, 0)); w += 4; }  /* :259-263 */

        if (flags & FL_INDEP) {                                     /* :265-270 */
            ib_len = dict_len;
            memcpy(&table, &template_table, sizeof table);
        } else if (ib_len > F_WINDOW) {                             /* :271-275 */
            size_t forget = ib_len - F_WINDOW;
            table.offset += forget;                                 /* mod.rs:72-74 */
            memmove(in_buffer, in_buffer + forget, F_WINDOW);
            ib_len = F_WINDOW;
        }
    }
    free(in_buffer);
    wr32le(out + w, 0); w += 4;                                     /* :277 */
    if (st->content_checksum) { wr32le(out + w, xxh32_digest(&content)); w += 4; }   /* :279-281 */
    *out_len = w;
    return LZFO_OK;
}

/* framed/decompress.rs:102-161 (header), :198-279 (decode_block), :284-288 (decompress_frame).
 * Returns the *inner* error kind (what decode_block / LZ4FrameReader::new return); note that
 * decompress_frame itself re-wraps block errors as InputError(io::Error::Other(..)) because it
 * goes through io::Read::read_to_end (:39-43, :286). */
int lzfo_frame_decompress(const uint8_t* in, size_t in_len, const uint8_t* dict, size_t dict_len,
                          uint8_t* out, size_t out_cap, size_t* out_len, size_t* consumed) {
    size_t r = 0, w = 0;
    int rc = LZFO_OK;
    uint8_t* window = NULL;
    *out_len = 0; if (consumed) *consumed = 0;
#define NEED(n) do { if (in_len - r < (size_t)(n)) { r = in_len; rc = LZFO_F_INPUT_ERROR; goto done; } } while (0)
    NEED(4); uint32_t magic = rd32le(in + r); r += 4;               /* :103 */
    if (magic != F_MAGIC) { rc = LZFO_F_WRONG_MAGIC; goto done; }   /* :104-106 */
    NEED(1); uint8_t flags_byte = in[r++];                          /* :108 */
    if ((flags_byte >> 6) != 1) { rc = LZFO_F_UNSUPPORTED_VERSION; goto done; }   /* header.rs:33-36 */
    if (flags_byte & 0x02) { rc = LZFO_F_RESERVED_FLAG_BITS; goto done; }         /* header.rs:37-39 */
    NEED(1); uint8_t bd = in[r++];                                  /* :110 */
    if (bd & 0x8F) { rc = LZFO_F_RESERVED_BD_BITS; goto done; }     /* header.rs:66-68 */
    size_t hstart = r - 2;
    if (flags_byte & FL_CSIZE) { NEED(8); r += 8; }                 /* :116-122 */
    if (flags_byte & FL_DICTID) { NEED(4); r += 4; }                /* :124-130 */
    NEED(1); uint8_t hc = in[r++];                                  /* :132 */
    if (hc != (uint8_t)(lzfo_xxh32(in + hstart, r - 1 - hstart, 0) >> 8)) { rc = LZFO_F_HEADER_CHECKSUM_FAIL; goto done; } /* :133-136 */
    unsigned size = (bd >> 4) & 7;                                  /* :153, header.rs:73-80 */
    if (size < 4) { rc = LZFO_F_UNIMPLEMENTED_BLOCKSIZE; goto done; }
    const size_t block_maxsize = (size_t)1 << (size * 2 + 8);

    xxh32_state content; xxh32_init(&content, 0);
    const int linked = !(flags_byte & FL_INDEP);
    size_t window_len = 0;                                          /* carryover_window :144-148 */
    if (linked) {
        window = (uint8_t*)malloc((dict_len > F_WINDOW ? dict_len : F_WINDOW) + F_WINDOW);
        if (!window) { rc = LZFO_OUT_CAPACITY; goto done; }
    }

    for (;;) {                                                      /* read_to_end loop */
        NEED(4); uint32_t block_length = rd32le(in + r); r += 4;    /* :205 */
        if (block_length == 0) {                                    /* :206-215 */
            if (flags_byte & FL_CSUM) {
                NEED(4); uint32_t c = rd32le(in + r); r += 4;
                if (c != xxh32_digest(&content)) { rc = LZFO_F_FRAME_CHECKSUM_FAIL; goto done; }
            }
            break;
        }
        int is_compressed = (block_length & F_INCOMPRESSIBLE) == 0; /* :217 */
        block_length &= ~F_INCOMPRESSIBLE;                          /* :218 */
        if (block_length > (uint32_t)block_maxsize) { rc = LZFO_F_BLOCK_SIZE_OVERFLOW; goto done; }   /* :220-222 */
        NEED(block_length); const uint8_t* buf = in + r; r += block_length;      /* :224-226 */
        if (flags_byte & FL_BLOCKSUM) {                             /* :228-235 */
            NEED(4); uint32_t c = rd32le(in + r); r += 4;
            if (c != lzfo_xxh32(buf, block_length, 0)) { rc = LZFO_F_BLOCK_CHECKSUM_FAIL; goto done; }
        }
        const uint8_t* prefix; size_t prefix_len;                   /* :238-245 */
        if (linked) {
            if (window_len == 0 && dict_len) { memcpy(window, dict, dict_len); window_len = dict_len; }   /* :239-241 */
            prefix = window; prefix_len = window_len;
        } else { prefix = dict; prefix_len = dict_len; }

        size_t olen = 0;                                            /* output: fresh empty Vec */
        if (is_compressed) {                                        /* :247-248 */
            int drc = lzfo_decompress_raw(buf, block_length, prefix, prefix_len, out + w, &olen, out_cap - w, block_maxsize);
            if (drc != LZFO_OK) { rc = drc; goto done; }
        } else {                                                    /* :249-251 */
            if (out_cap - w < block_length) { rc = LZFO_OUT_CAPACITY; goto done; }
            memcpy(out + w, buf, block_length); olen = block_length;
        }
        if (linked) {                                               /* :253-269 */
            if (olen < F_WINDOW) {
                size_t avail = window_len + olen;
                if (avail >= F_WINDOW) {                            /* :257-260 drain(..surplus) */
                    size_t surplus = avail - F_WINDOW;
                    memmove(window, window + surplus, window_len - surplus);
                    window_len -= surplus;
                }
                memcpy(window + window_len, out + w, olen); window_len += olen;   /* :261 */
            } else {                                                /* :262-266 */
                memcpy(window, out + w + olen - F_WINDOW, F_WINDOW); window_len = F_WINDOW;
            }
        }
        if (olen > block_maxsize) { rc = LZFO_F_BLOCK_SIZE_OVERFLOW; goto done; }   /* :272-274 */
        if (flags_byte & FL_CSUM) xxh32_update(&content, out + w, olen);            /* :276-278 */
        w += olen;
        /* io::Read adapter (:52-71) + read_to_end (:286): a block that decodes to zero bytes
         * makes read() return 0, which read_to_end takes for EOF -> Ok with what we have. */
        if (olen == 0) break;
    }
done:
    free(window);
    *out_len = w;
    if (consumed) *consumed = r;
    return rc;
#undef NEED
}
