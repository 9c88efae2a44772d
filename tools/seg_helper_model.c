// tools/seg_helper_model.c — round 6, review item 4: would HELPER wavefronts that place the off-chain matches of a block ahead of the
// resolver shorten the resolve stage of the segmented pipeline (lz4_decompress_seg.hip) by the 1.5 x that would justify building them?
//
// The resolve stage today (one resolver wavefront + one stager per block): per batch of 64 sequences the resolver runs one LDS round
// trip per DEPENDENCY LEVEL of the batch's matches (level 1 = every source byte is final when the batch starts — literals are placed by
// the records stage —, level k = copies from a level k - 1 match of the same batch).  Measured (profiles/r03_seg_pipeline_kernel_stats.txt,
// tools/round_microbench.hip): ~370 cycles per round all-in at ~13 active lanes, of which 184 are one ds_read + wait + ds_write and the
// rest grows with the active lanes; ~300 cycles of set-up per batch; the slowest text block 5.6-6.7 ms at one block per CU.
//
// The model replays every block's batches and prices the resolver's time as
//     T = sum over batches ( SETUP + sum over levels ( ROUND + LANE * lanes at that level ) )          [cycles]
// for the pipeline as it is, and for three ways helpers could take matches off the resolver's hands.  A helper can only place a match
// whose SOURCE IS FINAL when it works, and it must work through the block's LDS ring (the resolver's batches live there: the stager
// brings a granule in one sub-batch early and writes it back when the resolver is done, so a wavefront on another CU has no window in
// which its store to `out` would be seen) — i.e. helpers are extra wavefronts of the block's own workgroup running D batches ahead:
//     old(D)    helpers place every match of batch b whose source lies entirely in front of batch b - D (final for certain while
//               the resolver is still D batches behind); the resolver's levels are recomputed over what is left
//     level1    upper bound for "level-1 matches by helpers": every level-1 match of every batch is placed at no cost and in no time
//               (not realisable: a level-1 match whose source lies in batch b - 1 is final only when the resolver has finished b - 1)
//     oracle    lower bound of ANY scheme that keeps batches of 64: only the longest chain inside each batch is left (one lane per level)
// Output: per block kind and for the slowest block (the one that sets the launch's time at <= one block per CU) the predicted resolver
// time and the ratio to today's.
//   gcc -O2 -o /tmp/seg_helper_model tools/seg_helper_model.c oracle/lzf_oracle.c && /tmp/seg_helper_model corpus.bin
// ANALYSIS TOOL (links the oracle): not part of the product.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "../oracle/lzf_oracle.h"
#define BS (4u << 20)
#define B 64
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
static const double SETUP = 300.0, ROUND = 184.0, LANE = 14.3;      // 184 + 14.3 * 13 = 370 cycles for the measured average round
#define NV 6
static const char* VN[NV] = {"today", "old(1)", "old(2)", "old(4)", "level1", "oracle"};
int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); size_t total = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t* data = malloc(total); if (fread(data, 1, total, f) != total) return 1; fclose(f);
    size_t nblk = (total + BS - 1) / BS;
    seqs = malloc(sizeof(seq_t) * (BS / 2));
    uint8_t* comp = malloc(BS + 65536);
    double worst[NV] = {0}; size_t worst_blk = 0; double sumT[NV] = {0}, sumR[NV] = {0};
    double tot_matches = 0, tot_l1 = 0, tot_old[3] = {0, 0, 0}, tot_rounds = 0, tot_batches = 0, tot_src_prev = 0;
    printf("blk   nseq  batches rounds/batch  level-1  old(1) old(2) old(4) of matches |  resolver ms at 2.4 GHz: ");
    for (int v = 0; v < NV; ++v) printf("%8s", VN[v]);
    printf("\n");
    for (size_t b = 0; b < nblk; ++b) {
        size_t n = total - b * BS < BS ? total - b * BS : BS, clen = 0;
        lzfo_u32_table t; memset(&t, 0, sizeof t);
        if (lzfo_compress2(data + b * BS, n, 0, LZFO_TABLE_U32, &t, comp, n, &clen) != LZFO_OK) continue;
        parse(comp, clen);
        double T[NV] = {0}; double matches = 0, l1 = 0, old[3] = {0, 0, 0}, rounds = 0, batches = 0, src_prev = 0;
        for (size_t i0 = 0; i0 < nseq; i0 += B) {
            const size_t nb = nseq - i0 < B ? nseq - i0 : B;
            const uint32_t ob0 = seqs[i0].lo;
            // start of the output of batch b - D (D = 1, 2, 4); 0 when there is no such batch
            uint32_t before[3];
            for (int d = 0; d < 3; ++d) { const size_t D = (size_t)1 << d; before[d] = i0 >= D * B ? seqs[i0 - D * B].lo : 0u; }
            for (int v = 0; v < NV; ++v) {
                int lev[B]; int maxl = 0; int lanes[B + 2]; memset(lanes, 0, sizeof lanes);
                for (size_t j = 0; j < nb; ++j) {
                    const seq_t* s = &seqs[i0 + j]; lev[j] = 0;
                    if (!s->M) continue;
                    const uint32_t sp = s->M < s->off ? s->M : s->off; const int64_t s0 = (int64_t)s->mo - s->off, s1 = s0 + sp;
                    int placed = 0;
                    if (v >= 1 && v <= 3) placed = i0 >= ((size_t)1 << (v - 1)) * B && s1 <= (int64_t)before[v - 1];     // the source is final D batches early
                    int le = 0;
                    for (size_t i = 0; i < j; ++i) {
                        const seq_t* q = &seqs[i0 + i];
                        if (q->M && lev[i] > 0 && (int64_t)q->mo < s1 && (int64_t)q->mo + q->M > s0 && lev[i] > le) le = lev[i];
                    }
                    if (v == 0) {
                        matches += 1;
                        if (le == 0) { l1 += 1; if (s1 > (int64_t)before[0] && i0 >= B) src_prev += 1; }
                        for (int d = 0; d < 3; ++d) if (i0 >= ((size_t)1 << d) * B && s1 <= (int64_t)before[d]) old[d] += 1;
                    }
                    if (placed) { lev[j] = 0; continue; }                       // placed ahead: final when the batch starts, like a literal
                    lev[j] = le + 1;
                    if (lev[j] > maxl) maxl = lev[j];
                    lanes[lev[j]] += 1;
                }
                double tb = SETUP;
                if (v == 4) {          // today's levels; the level-1 round is the helpers' (free, instant): the resolver starts at level 2
                    for (int k = 2; k <= maxl; ++k) tb += ROUND + LANE * lanes[k];
                } else if (v == 5) {
                    int ml = 0; for (int k = 1; k <= B; ++k) if (lanes[k]) ml = k;
                    tb += ml * (ROUND + LANE);
                } else {
                    for (int k = 1; k <= maxl; ++k) tb += ROUND + LANE * lanes[k];
                }
                if (v == 0) { rounds += maxl; batches += 1; }
                sumR[v] += v == 4 ? (maxl > 1 ? maxl - 1 : 0) : maxl;
                T[v] += tb;
            }
        }
        printf("%3zu %6zu %7.0f %8.2f     %6.3f  %6.3f %6.3f %6.3f            |                          ", b, nseq, batches, rounds / batches, l1 / matches, old[0] / matches, old[1] / matches, old[2] / matches);
        for (int v = 0; v < NV; ++v) { printf("%8.2f", T[v] / 2.4e6); sumT[v] += T[v]; }
        printf("\n");
        if (T[0] > worst[0]) { memcpy(worst, T, sizeof T); worst_blk = b; }
        tot_matches += matches; tot_l1 += l1; tot_rounds += rounds; tot_batches += batches; tot_src_prev += src_prev;
        for (int d = 0; d < 3; ++d) tot_old[d] += old[d];
    }
    printf("\ncorpus: %.0f matches, level 1: %.3f (of those with a source that reaches into batch b - 1: %.3f of all matches), source final 1 / 2 / 4 batches early: %.3f / %.3f / %.3f; rounds per batch %.2f\n",
           tot_matches, tot_l1 / tot_matches, tot_src_prev / tot_matches, tot_old[0] / tot_matches, tot_old[1] / tot_matches, tot_old[2] / tot_matches, tot_rounds / tot_batches);
    printf("resolver rounds per batch:");
    for (int v = 0; v < NV; ++v) printf("  %s %.2f", VN[v], sumR[v] / tot_batches);
    printf("\n");
    printf("slowest block (%zu) — the launch's time at <= one block per CU — predicted resolver ms and speed-up over today:\n", worst_blk);
    for (int v = 0; v < NV; ++v) printf("  %-7s %6.2f ms  %.2f x\n", VN[v], worst[v] / 2.4e6, worst[0] / worst[v]);
    printf("sum over the corpus' blocks (throughput regime, four blocks per CU):\n");
    for (int v = 0; v < NV; ++v) printf("  %-7s %7.1f ms  %.2f x\n", VN[v], sumT[v] / 2.4e6, sumT[0] / sumT[v]);
    return 0;
}
