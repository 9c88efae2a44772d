// tools/round_rules.c — round 6: how many dependent LDS round trips does the copy stage's match phase need per batch, under which rule?
// A CPU model of lz4_decompress_batch_phase.inc's batching (up to 64 sequences, cut where the batch's output exceeds RING / 3) over the
// Silesia stand-in, block by block.  For the matches a lane moves on its own ("solo": not overlapping, <= 64 bytes, source inside the
// ring's intact history) it counts the rounds of
//   H      the kernel's rule: H = output start of the first unresolved match, every match whose source ends at or below H moves
//   exact  a match moves as soon as no unresolved match of the batch writes into its source (dependency levels)
//   range  a match moves as soon as every match between the sequence that produces its first source byte and the one that produces
//          its last is resolved (what two binary searches over the batch's output positions give a lane)
// and the other kinds (far: read back from HBM; cooperative: overlapping or long) per batch.
//   gcc -O2 -o /tmp/round_rules tools/round_rules.c oracle/lzf_oracle.c && /tmp/round_rules /tmp/silesia_mix.bin
// ANALYSIS TOOL (links the oracle): not part of the product.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "../oracle/lzf_oracle.h"
#define BS (4u << 20)
#define RING 4096u
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
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); size_t total = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t* data = malloc(total); if (fread(data, 1, total, f) != total) return 1; fclose(f);
    size_t nblk = (total + BS - 1) / BS;
    seqs = malloc(sizeof(seq_t) * (BS / 2));
    uint8_t* comp = malloc(BS + 65536);
    double batches = 0, nsq = 0, rH = 0, rE = 0, rR = 0, far = 0, coop = 0, solo = 0, nomatch = 0, lvl1 = 0, rH1 = 0;
    double hist[3][17] = {{0}};
    const uint32_t kSpanMax = RING / 3, kNearHist = RING - kSpanMax;
    for (size_t b = 0; b < nblk; ++b) {
        size_t n = total - b * BS < BS ? total - b * BS : BS, clen = 0;
        lzfo_u32_table t; memset(&t, 0, sizeof t);
        if (lzfo_compress2(data + b * BS, n, 0, LZFO_TABLE_U32, &t, comp, n, &clen) != LZFO_OK) continue;
        parse(comp, clen);
        size_t i0 = 0;
        while (i0 < nseq) {
            const uint32_t ob0 = seqs[i0].lo;
            size_t i1 = i0; uint32_t span = 0;
            while (i1 < nseq && i1 - i0 < 64) { const uint32_t tot = seqs[i1].L + seqs[i1].M; if (span + tot > kSpanMax) break; span += tot; ++i1; }
            if (i1 == i0) { ++i0; continue; }                     // a sequence larger than a batch: the solo path, no rounds
            const uint32_t near_lo = ob0 > kNearHist ? ob0 - kNearHist : 0u;
            const size_t nb = i1 - i0;
            int kind[64]; int lvE[64], lvR[64];                   // kind: 0 none, 1 solo, 2 far, 3 cooperative / slow
            for (size_t j = 0; j < nb; ++j) {
                const seq_t* s = &seqs[i0 + j];
                if (!s->M) { kind[j] = 0; ++nomatch; continue; }
                const uint32_t sp = s->M < s->off ? s->M : s->off, s0 = s->mo - s->off;
                if (s0 + sp <= near_lo) { kind[j] = 2; ++far; continue; }
                if (s0 >= near_lo && s->M <= s->off && s->M <= 64) { kind[j] = 1; ++solo; } else { kind[j] = 3; ++coop; }
            }
            // H rule over solo + cooperative (cooperative ones take a round of their own when they are the first unresolved)
            {
                uint8_t done[64]; int left = 0, r = 0;
                for (size_t j = 0; j < nb; ++j) { done[j] = !(kind[j] == 1 || kind[j] == 3); left += !done[j]; }
                while (left) {
                    size_t fI = 0; while (done[fI]) ++fI;
                    if (kind[fI] == 3) { done[fI] = 1; --left; ++r; continue; }
                    const uint32_t H = seqs[i0 + fI].mo; int moved = 0;
                    for (size_t j = fI; j < nb; ++j) if (!done[j] && kind[j] == 1) {
                        const seq_t* s = &seqs[i0 + j];
                        if (s->mo - s->off + s->M <= H) { done[j] = 2; ++moved; }
                    }
                    for (size_t j = 0; j < nb; ++j) if (done[j] == 2) done[j] = 1;
                    left -= moved; ++r;
                }
                rH += r; hist[0][r > 16 ? 16 : r] += 1;
            }
            // exact levels / range levels: level(j) = 1 + max level of the matches of the batch (any kind but far) that write into j's source
            int maxE = 0, maxR = 0;
            for (size_t j = 0; j < nb; ++j) {
                lvE[j] = lvR[j] = 0;
                if (kind[j] == 0 || kind[j] == 2) continue;
                const seq_t* s = &seqs[i0 + j];
                const uint32_t sp = s->M < s->off ? s->M : s->off; const int64_t s0 = (int64_t)s->mo - s->off, s1 = s0 + sp;
                int le = 0, lr = 0;
                for (size_t i = 0; i < j; ++i) {
                    const seq_t* q = &seqs[i0 + i];
                    if (kind[i] == 0) { /* literals only */ }
                    // exact: i's match destination [mo, mo + M) overlaps [s0, s1)
                    if (kind[i] != 0 && kind[i] != 2 && (int64_t)q->mo < s1 && (int64_t)q->mo + q->M > s0 && lvE[i] > le) le = lvE[i];
                    // range: i's whole output [lo, mo + M) overlaps [s0, s1)   (sequence-granular)
                    if ((int64_t)q->lo < s1 && (int64_t)q->mo + q->M > s0 && lvR[i] > lr) lr = lvR[i];
                }
                // far matches written by this batch count too (they land in the ring before the rounds): no dependency
                lvE[j] = le + 1; lvR[j] = lr + 1;
                if (lvE[j] > maxE) maxE = lvE[j]; if (lvR[j] > maxR) maxR = lvR[j];
                if (lvE[j] == 1) ++lvl1;
            }
            rE += maxE; rR += maxR; hist[1][maxE > 16 ? 16 : maxE] += 1; hist[2][maxR > 16 ? 16 : maxR] += 1;
            batches += 1; nsq += nb; i0 = i1;
        }
    }
    printf("batches %.0f, sequences %.0f (%.1f per batch): no match %.3f, solo %.3f, far %.3f, cooperative %.3f per sequence\n", batches, nsq, nsq / batches, nomatch / nsq, solo / nsq, far / nsq, coop / nsq);
    printf("rounds per batch: H rule %.2f | exact levels %.2f | range levels %.2f ; matches at exact level 1: %.3f of solo + cooperative\n", rH / batches, rE / batches, rR / batches, lvl1 / (solo + coop));
    for (int k = 0; k < 3; ++k) { printf("%s:", k == 0 ? "H    " : k == 1 ? "exact" : "range"); for (int r = 0; r <= 16; ++r) printf(" %5.1f%%", 100.0 * hist[k][r] / batches); printf("\n"); }
    return 0;
}
