// tools/seg_rounds.c — design input for the resolver of the segmented decompress pipeline: what the rounds of a block are made of.
// Per 4 MiB block of the Silesia stand-in: batches of 64 sequences, dependency levels inside a batch, and per level the lanes by
// the resolver's copy paths (two-ended classes 1..4; run-length; long in-lane; doubling; whole wave).
//   gcc -O2 -o /tmp/seg_rounds tools/seg_rounds.c oracle/lzf_oracle.c && /tmp/seg_rounds corpus.bin
// ANALYSIS TOOL (links the oracle): not part of the product.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "../oracle/lzf_oracle.h"
#define BS (4u<<20)
typedef struct { uint32_t L, M, off, lo, mo; } seq_t;
static seq_t* seqs; static size_t nseq;
static void parse(const uint8_t* c, size_t len) {
    size_t p = 0; uint32_t o = 0; nseq = 0;
    while (p < len) {
        seq_t s; uint8_t tok = c[p++]; uint32_t L = tok >> 4;
        if (L == 15) { uint8_t b; do { b = c[p++]; L += b; } while (b == 255); }
        s.L = L; s.lo = o; p += L; o += L; s.mo = o;
        if (len - p < 2) { s.M = 0; s.off = 0; seqs[nseq++] = s; break; }
        s.off = c[p] | (c[p + 1] << 8); p += 2;
        uint32_t M = tok & 15;
        if (M == 15) { uint8_t b; do { b = c[p++]; M += b; } while (b == 255); }
        M += 4; s.M = M; o += M; seqs[nseq++] = s;
    }
}
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); size_t total = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t* data = malloc(total); if (fread(data, 1, total, f) != total) return 1; fclose(f);
    uint8_t* comp = malloc(BS + 65536); seqs = malloc(sizeof(seq_t) * (BS / 2));
    uint8_t* lev = malloc(BS + 16);
    printf("blk   nseq batches rounds | rounds with: only-fast  any-slow | slow lanes: rle lng dbl wave | wave lanes per round 1 2 3-4 5-8 >8 | wave bytes  mean M | lng+wave per round hist 1 2 3-4 5-8 >8 | rle lanes by M: 4-7 8-16 17-32 33-64 >64\n");
    for (size_t b0 = 0, bi = 0; b0 < total; b0 += BS, ++bi) {
        size_t n = total - b0 < BS ? total - b0 : BS, clen = 0;
        lzfo_u32_table t; memset(&t, 0, sizeof t);
        if (lzfo_compress2(data + b0, n, 0, LZFO_TABLE_U32, &t, comp, n, &clen) != LZFO_OK) { printf("%zu stored\n", bi); continue; }
        parse(comp, clen);
        unsigned long rh[5] = {0}, rounds = 0, fastonly = 0, anyslow = 0, n_rle = 0, n_lng = 0, n_dbl = 0, n_wave = 0, wbytes = 0, wh[5] = {0}, lh[5] = {0};
        for (size_t i0 = 0; i0 < nseq; i0 += 64) {
            size_t i1 = i0 + 64 < nseq ? i0 + 64 : nseq; uint32_t ts = seqs[i0].lo; int lv[64], mx = 0;
            for (size_t j = i0; j < i1; ++j) {
                seq_t* s = &seqs[j]; memset(lev + s->lo, 0, s->L); int d = 0;
                if (s->M) {
                    uint32_t span = s->M < s->off ? s->M : s->off; int64_t a = (int64_t)s->mo - s->off; int m = 0;
                    for (uint32_t k = 0; k < span; ++k) { int64_t q = a + k; if (q >= (int64_t)ts && lev[q] > m) m = lev[q]; }
                    d = m + 1; memset(lev + s->mo, d, s->M);
                }
                lv[j - i0] = d; if (d > mx) mx = d;
            }
            for (int l = 1; l <= mx; ++l) {
                int fast = 0, rle = 0, lng = 0, dbl = 0, wave = 0;
                for (size_t j = i0; j < i1; ++j) if (lv[j - i0] == l) {
                    seq_t* s = &seqs[j];
                    if (s->M <= 64 && s->off >= s->M) ++fast;
                    else if (s->off == 1 || s->off == 2 || s->off == 4) { ++rle; ++rh[s->M < 8 ? 0 : s->M <= 16 ? 1 : s->M <= 32 ? 2 : s->M <= 64 ? 3 : 4]; }
                    else if (s->off >= s->M && s->M <= 160) ++lng;
                    else if (s->M <= 64) ++dbl;
                    else { ++wave; wbytes += s->M; }
                }
                ++rounds; if (rle + lng + dbl + wave) ++anyslow; else ++fastonly;
                n_rle += rle; n_lng += lng; n_dbl += dbl; n_wave += wave;
                if (wave) ++wh[wave == 1 ? 0 : wave == 2 ? 1 : wave <= 4 ? 2 : wave <= 8 ? 3 : 4];
                int lw = lng + wave; if (lw) ++lh[lw == 1 ? 0 : lw == 2 ? 1 : lw <= 4 ? 2 : lw <= 8 ? 3 : 4];
            }
        }
        printf("%2zu %7zu %6zu %6lu | %6lu %6lu | %6lu %6lu %6lu %6lu | %5lu %5lu %5lu %5lu %5lu | %8lu %6.0f | %5lu %5lu %5lu %5lu %5lu | rle M: %lu %lu %lu %lu %lu\n", bi, nseq, (nseq + 63) / 64, rounds, fastonly, anyslow,
               n_rle, n_lng, n_dbl, n_wave, wh[0], wh[1], wh[2], wh[3], wh[4], wbytes, n_wave ? (double)wbytes / n_wave : 0.0, lh[0], lh[1], lh[2], lh[3], lh[4], rh[0], rh[1], rh[2], rh[3], rh[4]);
    }
    return 0;
}
