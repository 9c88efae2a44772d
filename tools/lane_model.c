// tools/lane_model.c — round 4, review item 3: would SEVERAL BLOCKS PER WAVEFRONT (a group of G = 32 / 16 / 8 lanes per block instead of
// 64) lower the wave-instructions per sequence of the large-batch decompress kernel?  A CPU model of the pair kernel's two stages on
// the Silesia stand-in, block by block:
//   COPY   batches of G sequences per block; rounds per batch by the kernel's own rule (H = output start of the first unresolved
//          match; every match whose source ends at or below H moves in this round); matches moved per round = active lanes;
//          a wave runs 64 / G blocks in lock-step, so it executes max(rounds) over their batches
//   PARSE  chunks of G regions of S bytes per block; passes of the fixed point "entry[i + 1] = exit of region i" until nothing changes
//          (region 0 starts on a true token), again max over the 64 / G blocks of a wave
// and the prediction: instructions per sequence = measured per-stage costs of the G = 64 kernel (profiles/r01_decompress_phase_counters.txt:
// parse 13.0 of which the fixed point ~ passes x 40 instructions per chunk, rounds 7.4 = 79 instructions per round, set-up 4.5,
// literals 2.0, far matches 1.3, flush 0.8) with the round and pass counts replaced by the model's.
//   gcc -O2 -o /tmp/lane_model tools/lane_model.c oracle/lzf_oracle.c && /tmp/lane_model corpus.bin
// ANALYSIS TOOL (links the oracle): not part of the product.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "../oracle/lzf_oracle.h"
#define BS (4u << 20)
typedef struct { uint32_t L, M, off, lo, mo, tok; } seq_t;      // lo / mo: output position of the literals / of the match; tok: input position of the token
static seq_t* seqs; static size_t nseq;
static void parse(const uint8_t* c, size_t len) {
    size_t p = 0; uint32_t o = 0; nseq = 0;
    while (p < len) {
        seq_t s; s.tok = (uint32_t)p; uint8_t tok = c[p++]; uint32_t L = tok >> 4;
        if (L == 15) { uint8_t b; do { b = c[p++]; L += b; } while (b == 255); }
        s.L = L; s.lo = o; p += L; o += L; s.mo = o;
        if (len - p < 2) { s.M = 0; s.off = 0; seqs[nseq++] = s; break; }
        s.off = c[p] | (c[p + 1] << 8); p += 2;
        uint32_t M = tok & 15;
        if (M == 15) { uint8_t b; do { b = c[p++]; M += b; } while (b == 255); }
        M += 4; s.M = M; o += M; seqs[nseq++] = s;
    }
}
// rounds of one batch [i0, i1) by the H rule; act[] gets the matches moved per round
static int batch_rounds(size_t i0, size_t i1, int* act, int maxr) {
    uint8_t done[64]; int left = 0, r = 0;
    for (size_t j = i0; j < i1; ++j) { done[j - i0] = seqs[j].M == 0; left += !done[j - i0]; }
    while (left) {
        size_t f = i0; while (done[f - i0]) ++f;
        const uint32_t H = seqs[f].mo; int moved = 0;
        for (size_t j = f; j < i1; ++j) if (!done[j - i0]) {
            const seq_t* s = &seqs[j];
            const int64_t src_end = (int64_t)s->mo - s->off + (s->M < s->off ? s->M : s->off);
            if (j == f || src_end <= (int64_t)H) { done[j - i0] = 2; ++moved; }
        }
        for (size_t j = i0; j < i1; ++j) if (done[j - i0] == 2) done[j - i0] = 1;
        left -= moved; if (r < maxr) act[r] = moved; ++r;
    }
    return r;
}
// token chain stepping: next token position at or after position p when a token starts at p (walks one token)
static uint32_t next_tok(const uint8_t* c, size_t len, uint32_t p) {
    if (p >= len) return (uint32_t)len;
    uint8_t tok = c[p++]; uint32_t L = tok >> 4;
    if (L == 15) { uint8_t b; do { if (p >= len) return (uint32_t)len; b = c[p++]; L += b; } while (b == 255); }
    p += L; if (len < p || len - p < 2) return (uint32_t)len;
    p += 2;
    if ((tok & 15) == 15) { uint8_t b; do { if (p >= len) return (uint32_t)len; b = c[p++]; } while (b == 255); }
    return p;
}
// passes of the region fixed point over one chunk of G regions of S bytes that starts on the true token t0
static int chunk_passes(const uint8_t* c, size_t len, uint32_t t0, int G, int S, uint32_t* t_end) {
    uint32_t entry[64], base = t0;
    for (int i = 0; i < G; ++i) entry[i] = base + (uint32_t)i * S;          // guess: a token starts with the region (region 0: true)
    int passes = 0, changed = 1;
    while (changed) {
        changed = 0; ++passes;
        uint32_t ex[64];
        for (int i = 0; i < G; ++i) { uint32_t p = entry[i], end = base + (uint32_t)(i + 1) * S; while (p < end && p < len) p = next_tok(c, len, p); ex[i] = p; }
        uint32_t mx = 0;
        for (int i = 0; i + 1 < G; ++i) { if (ex[i] > mx) mx = ex[i]; uint32_t e = mx > base + (uint32_t)(i + 1) * S ? mx : base + (uint32_t)(i + 1) * S;
            // entry of region i + 1 = the furthest exit of the regions before it (prefix max), not before its own start
            if (e != entry[i + 1]) { entry[i + 1] = e; changed = 1; } }
        if (passes > 200) break;
        if (!changed) { uint32_t m2 = 0; for (int i = 0; i < G; ++i) if (ex[i] > m2) m2 = ex[i]; *t_end = m2; }
    }
    return passes;
}
int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); size_t total = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t* data = malloc(total); if (fread(data, 1, total, f) != total) return 1; fclose(f);
    size_t nblk = (total + BS - 1) / BS;
    uint8_t** comps = calloc(nblk, sizeof *comps); size_t* clens = calloc(nblk, sizeof *clens);
    seq_t** bseq = calloc(nblk, sizeof *bseq); size_t* bn = calloc(nblk, sizeof *bn);
    seqs = malloc(sizeof(seq_t) * (BS / 2));
    size_t ok = 0, totseq = 0;
    for (size_t b = 0; b < nblk; ++b) {
        size_t n = total - b * BS < BS ? total - b * BS : BS, clen = 0;
        uint8_t* comp = malloc(BS + 65536);
        lzfo_u32_table t; memset(&t, 0, sizeof t);
        if (lzfo_compress2(data + b * BS, n, 0, LZFO_TABLE_U32, &t, comp, n, &clen) != LZFO_OK) { free(comp); continue; }
        parse(comp, clen);
        comps[ok] = comp; clens[ok] = clen; bseq[ok] = malloc(sizeof(seq_t) * nseq); memcpy(bseq[ok], seqs, sizeof(seq_t) * nseq); bn[ok] = nseq; totseq += nseq; ++ok;
    }
    printf("# %zu compressible 4 MiB blocks, %zu sequences\n", ok, totseq);
    const int S = 24;                                     // region bytes of the headline kernel (paired24)
    printf("# G = lanes per block | COPY: rounds per batch of G sequences (one block alone), the same as max over the 64/G blocks a wave runs in lock-step,\n");
    printf("#   rounds per 64 sequences, matches moved per round (= active lanes of the round's DS instructions, of 64) | PARSE (S = %d): passes per chunk alone, max over 64/G blocks\n", S);
    printf("  G  rounds/batch  max over wave  rounds per 64 seq  active lanes/round  | passes/chunk  max over wave | predicted wave-instructions per sequence (today: 29.0 measured at G = 64)\n");
    for (int G = 64; G >= 8; G >>= 1) {
        const int W = 64 / G;                            // blocks per wave
        double sum_r = 0, sum_b = 0, sum_rw = 0, sum_bw = 0, sum_act = 0, sum_rounds = 0;
        // COPY: blocks taken W at a time (k, k + 1, ... — neighbours in the corpus order, i.e. of similar kind: the favourable case), batch index by batch index
        for (size_t k = 0; k + W <= ok; k += W) {
            size_t nb_max = 0; for (int w = 0; w < W; ++w) { size_t nb = (bn[k + w] + G - 1) / G; if (nb > nb_max) nb_max = nb; }
            for (size_t bi = 0; bi < nb_max; ++bi) {
                int rmax = 0;
                for (int w = 0; w < W; ++w) {
                    seqs = bseq[k + w]; nseq = bn[k + w];
                    size_t i0 = bi * G; if (i0 >= nseq) continue;
                    size_t i1 = i0 + G < nseq ? i0 + G : nseq; int act[64];
                    int r = batch_rounds(i0, i1, act, 64);
                    sum_r += r; sum_b += 1; if (r > rmax) rmax = r;
                    for (int q = 0; q < r && q < 64; ++q) { sum_act += act[q]; sum_rounds += 1; }
                }
                sum_rw += rmax; sum_bw += 1;
            }
        }
        // PARSE: chunks of G regions, W blocks in lock-step (chunk index by chunk index)
        double sum_p = 0, sum_c = 0, sum_pw = 0, sum_cw = 0;
        for (size_t k = 0; k + W <= ok; k += W) {
            uint32_t pos[8] = {0}; int live = W;
            while (live) {
                int pmax = 0; live = 0;
                for (int w = 0; w < W; ++w) if (pos[w] < clens[k + w]) {
                    uint32_t e = pos[w]; int p = chunk_passes(comps[k + w], clens[k + w], pos[w], G, S, &e);
                    if (e <= pos[w]) e = (uint32_t)clens[k + w];
                    pos[w] = e; sum_p += p; sum_c += 1; if (p > pmax) pmax = p; ++live;
                }
                if (live) { sum_pw += pmax; sum_cw += 1; }
            }
        }
        const double rpb = sum_r / sum_b, rpw = sum_rw / sum_bw, r64 = rpw;            // a wave batch covers W x G = 64 sequences
        const double ppc = sum_p / sum_c, ppw = sum_pw / sum_cw;
        // prediction.  Measured at G = 64 (29.0 in all): parse 13.0 = fixed point (passes x 40 instructions per chunk) + the rest; rounds 7.4; set-up 4.5; literals 2.0; far 1.3; flush 0.8
        static double p64 = 0, r64_0 = 0;
        if (G == 64) { p64 = ppw; r64_0 = r64; }
        const double seq_per_chunk = (double)totseq / (sum_c > 0 ? (sum_c * G / 64.0) : 1);      // sequences per 64 regions
        const double fix64 = p64 * 40.0 / seq_per_chunk, rest = 13.0 - fix64;
        const double parse = rest + ppw * 40.0 / seq_per_chunk;
        const double rounds = 7.4 * r64 / r64_0;
        printf("%3d  %11.2f  %13.2f  %17.2f  %18.2f  | %12.2f  %13.2f | parse %.1f + rounds %.1f + set-up 4.5 + literals 2.0 + far 1.3 + flush 0.8 = %.1f\n",
               G, rpb, rpw, r64, sum_act / sum_rounds, ppc, ppw, parse, rounds, parse + rounds + 4.5 + 2.0 + 1.3 + 0.8);
    }
    return 0;
}
