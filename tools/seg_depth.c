// tools/seg_depth.c — design input for the segmented decompress pipeline: the depth of the dependency DAG of a block's matches
// ("match j reads bytes match i wrote"), for the whole 4 MiB block and for windows of 64 ... 65536 sequences / 16 ... 256 KiB of
// output (everything in front of a window counted as final).  The depth is the number of LDS round trips no decoder can avoid.
//   gcc -O2 -o /tmp/seg_depth tools/seg_depth.c oracle/lzf_oracle.c && /tmp/seg_depth corpus.bin
// (corpus.bin: rust-lz-fear_amd/synth.py silesia_mix() written to a file).  ANALYSIS TOOL (links the oracle): not part of the product.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "../oracle/lzf_oracle.h"
#define BS (4u<<20)
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
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); size_t total = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t* data = malloc(total); if (fread(data, 1, total, f) != total) return 1; fclose(f);
    uint8_t* comp = malloc(BS + 65536); seqs = malloc(sizeof(seq_t) * (BS / 2));
    uint32_t* lev = malloc(4 * (BS + 16));
    uint32_t* slev = malloc(4 * (BS / 2));
    int tiles[] = {64, 256, 1024, 4096, 16384, 65536, 1 << 30};
    printf("blk nseq  depth(whole) | mean max depth per tile of T seqs: 64 256 1024 4096 16384 65536 | by output tile bytes 16K 64K 256K\n");
    for (size_t b0 = 0, bi = 0; b0 < total; b0 += BS, ++bi) {
        size_t n = total - b0 < BS ? total - b0 : BS, clen = 0;
        lzfo_u32_table t; memset(&t, 0, sizeof t);
        int st = lzfo_compress2(data + b0, n, 0, LZFO_TABLE_U32, &t, comp, n, &clen);
        if (st != LZFO_OK) { printf("%zu stored\n", bi); continue; }
        parse(comp, clen);
        printf("%2zu %6zu ", bi, nseq);
        for (int ti = 6; ti >= 0; --ti) {
            int T = tiles[ti]; double sum = 0; int cnt = 0; uint32_t whole = 0;
            for (size_t i0 = 0; i0 < nseq; i0 += T) {
                size_t i1 = i0 + T < nseq ? i0 + T : nseq; uint32_t ts = seqs[i0].lo, mx = 0;
                for (size_t j = i0; j < i1; ++j) {
                    seq_t* s = &seqs[j];
                    for (uint32_t k = 0; k < s->L; ++k) lev[s->lo + k] = 0;
                    uint32_t d = 0;
                    if (s->M) {
                        uint32_t span = s->M < s->off ? s->M : s->off;
                        int64_t a = (int64_t)s->mo - s->off;
                        uint32_t m = 0;
                        for (uint32_t k = 0; k < span; ++k) { int64_t q = a + k; if (q >= (int64_t)ts && lev[q] > m) m = lev[q]; }
                        d = m + 1;
                        for (uint32_t k = 0; k < s->M; ++k) lev[s->mo + k] = d;
                    }
                    slev[j] = d;
                    if (d > mx) mx = d;
                }
                sum += mx; cnt++; if (mx > whole) whole = mx;
            }
            if (ti == 6) {
                printf("%6u | ", whole);
                // level histogram summary: how many sequences at levels; print percentiles of level
                // parallelism: nseq / depth
            } 
            else printf("%7.1f ", sum / cnt);
            if (ti == 6) { /* reorder print later */ }
        }
        // output-byte tiles
        int bts[] = {16384, 65536, 262144};
        printf("| ");
        for (int ti = 0; ti < 3; ++ti) {
            uint32_t TB = bts[ti]; double sum = 0; int cnt = 0;
            size_t j = 0;
            while (j < nseq) {
                uint32_t ts = seqs[j].lo, mx = 0;
                while (j < nseq && seqs[j].lo < ts + TB) {
                    seq_t* s = &seqs[j];
                    for (uint32_t k = 0; k < s->L; ++k) lev[s->lo + k] = 0;
                    uint32_t d = 0;
                    if (s->M) {
                        uint32_t span = s->M < s->off ? s->M : s->off;
                        int64_t a = (int64_t)s->mo - s->off; uint32_t m = 0;
                        for (uint32_t k = 0; k < span; ++k) { int64_t q = a + k; if (q >= (int64_t)ts && lev[q] > m) m = lev[q]; }
                        d = m + 1;
                        for (uint32_t k = 0; k < s->M; ++k) lev[s->mo + k] = d;
                    }
                    if (d > mx) mx = d; ++j;
                }
                sum += mx; cnt++;
            }
            printf("%7.1f ", sum / cnt);
        }
        printf("\n");
    }
    return 0;
}
