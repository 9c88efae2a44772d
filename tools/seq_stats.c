// tools/seq_stats.c — analysis of the LZ4 sequence structure of a corpus (design input for the decompress kernel):
// per 4 MiB block, compressed with the oracle: match-distance histogram, literal / match length classes,
// dependency depth of a 64-sequence batch under "all lanes copy, repeat until nothing changes",
// and how quickly a token walk from an arbitrary byte re-synchronises with the true chain.
//   gcc -O2 -o /tmp/seq_stats tools/seq_stats.c oracle/lzf_oracle.c && /tmp/seq_stats corpus.bin
// TEST/ANALYSIS TOOL (links the oracle): not part of the product.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "../oracle/lzf_oracle.h"

#define BS (4u << 20)
typedef struct { uint32_t pos, src, L, M, off, lo, mo; } seq_t;

static seq_t* seqs; static size_t nseq;

static void parse(const uint8_t* c, size_t len) {
    size_t p = 0; uint32_t o = 0; nseq = 0;
    while (p < len) {
        seq_t s; s.pos = (uint32_t)p;
        uint8_t tok = c[p++];
        uint32_t L = tok >> 4;
        if (L == 15) { uint8_t b; do { b = c[p++]; L += b; } while (b == 255); }
        s.src = (uint32_t)p; s.L = L; s.lo = o; p += L; o += L; s.mo = o;
        if (len - p < 2) { s.M = 0; s.off = 0; seqs[nseq++] = s; break; }
        s.off = c[p] | (c[p + 1] << 8); p += 2;
        uint32_t M = tok & 15;
        if (M == 15) { uint8_t b; do { b = c[p++]; M += b; } while (b == 255); }
        M += 4; s.M = M; o += M;
        seqs[nseq++] = s;
    }
}
// next token position for a walk from arbitrary p (garbage tolerant)
static size_t next_tok(const uint8_t* c, size_t len, size_t p) {
    uint8_t tok = c[p++];
    uint32_t L = tok >> 4;
    if (L == 15) { uint8_t b; do { if (p >= len) return len; b = c[p++]; L += b; } while (b == 255); }
    p += L; if (p + 2 > len) return len; p += 2;
    if ((tok & 15) == 15) { uint8_t b; do { if (p >= len) return len; b = c[p++]; } while (b == 255); }
    return p;
}

int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); size_t total = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t* data = malloc(total); if (fread(data, 1, total, f) != total) return 1; fclose(f);
    uint8_t* comp = malloc(BS + 65536); seqs = malloc(sizeof(seq_t) * (BS / 2));
    uint8_t* istok = malloc(BS + 65536);
    double dist_hist[18] = {0}; double Lc[8] = {0}, Mc[8] = {0}; double tseq = 0, tout = 0, tcomp = 0;
    double depth_hist[40] = {0}, nbatch = 0; double span_sum = 0;
    double sync_tok_hist[40] = {0}, sync_byte_hist[20] = {0}, nsync = 0;
    double rmax64 = 0, rmax128 = 0, rcnt64 = 0, rcnt128 = 0, rtok64 = 0, rtok128 = 0;
    int only = argc > 2 ? atoi(argv[2]) : -1;
    for (size_t b0 = 0, bi = 0; b0 < total; b0 += BS, ++bi) {
        if (only >= 0 && (int)bi != only) continue;
        size_t n = total - b0 < BS ? total - b0 : BS, clen = 0;
        lzfo_u32_table t; memset(&t, 0, sizeof t);
        int st = lzfo_compress2(data + b0, n, 0, LZFO_TABLE_U32, &t, comp, n, &clen);
        if (st != LZFO_OK) { printf("block %zu stored\n", bi); continue; }
        parse(comp, clen);
        tseq += nseq; tout += n; tcomp += clen;
        memset(istok, 0, clen + 1);
        for (size_t i = 0; i < nseq; ++i) istok[seqs[i].pos] = 1;
        for (size_t i = 0; i < nseq; ++i) {
            seq_t* s = &seqs[i];
            int lc = s->L == 0 ? 0 : s->L <= 4 ? 1 : s->L <= 8 ? 2 : s->L <= 16 ? 3 : s->L <= 32 ? 4 : s->L <= 64 ? 5 : s->L <= 256 ? 6 : 7;
            Lc[lc]++;
            if (s->M) {
                int mc = s->M <= 7 ? 0 : s->M <= 16 ? 1 : s->M <= 32 ? 2 : s->M <= 64 ? 3 : s->M <= 256 ? 4 : s->M <= 1024 ? 5 : 6;
                Mc[mc]++;
                int h = 0; while ((1u << h) < s->off) ++h; dist_hist[h]++;
                if (s->off < s->M) Mc[7]++;   // overlapping
            }
        }
        // batches of 64 sequences (span <= 1365 bytes like RING/3 .. use 64 seqs or 4096 bytes): dependency depth.
        // depth[j] = 1 + max depth of matches i<j in the batch whose dest intersects src range of j; literals depth 0
        for (size_t i0 = 0; i0 < nseq;) {
            size_t i1 = i0; uint32_t ob0 = seqs[i0].lo;
            while (i1 < nseq && i1 - i0 < 64 && seqs[i1].mo + seqs[i1].M - ob0 <= 5400) ++i1;
            if (i1 == i0) { ++i0; continue; }
            int depth[64], maxd = 0;
            for (size_t j = i0; j < i1; ++j) {
                seq_t* s = &seqs[j]; int d = 0;
                if (s->M) {
                    d = 1;
                    uint32_t span = s->M < s->off ? s->M : s->off;
                    uint32_t a = s->mo - s->off, e = a + span;
                    if (s->off < s->M && s->off >= 8) d += (s->M + s->off - 1) / s->off - 1;   // self-overlap via pieces
                    for (size_t i = i0; i < j; ++i) {
                        seq_t* r = &seqs[i];
                        if (r->M && r->mo < e && r->mo + r->M > a) { int dd = depth[i - i0] + 1; if (s->off < s->M && s->off >= 8) dd += (s->M + s->off - 1) / s->off - 1; if (dd > d) d = dd; }
                    }
                }
                depth[j - i0] = d; if (d > maxd) maxd = d;
            }
            depth_hist[maxd > 39 ? 39 : maxd]++; nbatch++; span_sum += seqs[i1 - 1].mo + seqs[i1 - 1].M - ob0;
            i0 = i1;
        }
        // re-synchronisation: from every 64th byte, walk until hitting a true token
        for (size_t p0 = 64; p0 + 64 < clen; p0 += 64) {
            size_t p = p0; int k = 0;
            while (p < clen && !istok[p] && k < 39) { p = next_tok(comp, clen, p); ++k; }
            sync_tok_hist[k]++; nsync++;
            size_t d = p - p0; int h = 0; while ((1u << h) < d + 1) ++h; sync_byte_hist[h > 19 ? 19 : h]++;
        }
        // tokens per region of 64 / 128 bytes: mean and mean-of-max over 64 consecutive regions
        for (int S = 64; S <= 128; S *= 2) {
            size_t nreg = clen / S; uint16_t* cnt = calloc(nreg + 1, 2);
            for (size_t i = 0; i < nseq; ++i) if (seqs[i].pos / S < nreg) cnt[seqs[i].pos / S]++;
            for (size_t r = 0; r + 64 <= nreg; r += 64) {
                int mx = 0, sm = 0; for (int k = 0; k < 64; ++k) { if (cnt[r + k] > mx) mx = cnt[r + k]; sm += cnt[r + k]; }
                if (S == 64) { rmax64 += mx; rtok64 += sm; rcnt64++; } else { rmax128 += mx; rtok128 += sm; rcnt128++; }
            }
            free(cnt);
        }
    }
    printf("sequences %.0f, out bytes/seq %.2f, comp bytes/seq %.2f, ratio %.3f\n", tseq, tout / tseq, tcomp / tseq, tout / tcomp);
    printf("literal length classes 0 | 1-4 | 5-8 | 9-16 | 17-32 | 33-64 | 65-256 | >256 (%% of sequences)\n ");
    for (int i = 0; i < 8; ++i) printf(" %.2f", 100 * Lc[i] / tseq); printf("\n");
    printf("match length classes 4-7 | 8-16 | 17-32 | 33-64 | 65-256 | 257-1024 | >1024 | overlapping\n ");
    for (int i = 0; i < 8; ++i) printf(" %.2f", 100 * Mc[i] / tseq); printf("\n");
    printf("match distance: cumulative %% with offset <= 2^h\n ");
    double c = 0; for (int h = 0; h <= 16; ++h) { c += dist_hist[h]; printf(" %d:%.1f", h, 100 * c / tseq); } printf("\n");
    printf("batch (64 seq, span<=5400) dependency depth histogram (%% of batches), mean span %.0f B\n ", span_sum / nbatch);
    double md = 0; for (int d = 0; d < 40; ++d) { md += d * depth_hist[d]; if (depth_hist[d] > 0) printf(" %d:%.1f", d, 100 * depth_hist[d] / nbatch); } printf("\n  mean depth %.2f\n", md / nbatch);
    printf("resync from arbitrary byte: tokens walked until on the true chain (%% of starts)\n ");
    c = 0; for (int k = 0; k < 40; ++k) { c += sync_tok_hist[k]; if (sync_tok_hist[k] > 0) printf(" %d:%.1f", k, 100 * c / nsync); } printf("\n bytes (<=2^h):");
    c = 0; for (int h = 0; h < 20; ++h) { c += sync_byte_hist[h]; if (sync_byte_hist[h] > 0) printf(" %d:%.1f", h, 100 * c / nsync); } printf("\n");
    printf("tokens per 64-byte region: mean %.2f, mean of max over 64 regions %.2f; 128-byte: %.2f / %.2f\n",
           rtok64 / rcnt64 / 64, rmax64 / rcnt64, rtok128 / rcnt128 / 64, rmax128 / rcnt128);
    return 0;
}
