// tools/compress_speculation_probe.c — can ONE block be COMPRESSED by several wavefronts, bit-exactly, by speculation?
// Does a parse started W bytes early with an empty table reproduce the true parse (sequence starts + live table entries)
// at a segment boundary?  Minimal restatement of compress2's search loop (mod.rs:165-238), no output.  Answer on the Silesia
// stand-in: no — see profiles/r03_seg_design_statistics.txt.  gcc -O2 -o /tmp/spec tools/compress_speculation_probe.c && /tmp/spec corpus.bin SEG W
// ANALYSIS TOOL: not part of the product.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#define BS (4u<<20)
static inline uint32_t hash5(const uint8_t* p, size_t rem) { uint64_t v = 0; if (rem >= 8) memcpy(&v, p, 8); return (uint32_t)(((v << 24) * 889523592379ull) >> 52); }
typedef struct { uint32_t t[4096]; } tab_t;
// parse input[cursor..len) from table state T; stops at the first sequence start >= stop (or end). Records sequence starts >= rec_from into seqs.
// snapshot: when the first sequence start >= snap_at is reached, copy table to *snap and set *snap_pos.
static size_t parse(const uint8_t* in, size_t len, size_t cursor, tab_t* T, size_t stop, size_t snap_at, tab_t* snap, size_t* snap_pos, int* fullmatch) {
    size_t init = cursor; int snapped = 0;
    while (cursor < len) {
        size_t ls = cursor;
        if (!snapped && ls >= snap_at) { *snap = *T; *snap_pos = ls; snapped = 1; }
        if (ls >= stop) return ls;
        size_t sc = 64, step = 1; 
        for (;;) {
            if ((len > cursor ? len - cursor : 0) < 12) return len;
            uint32_t h = hash5(in + cursor, len - cursor);
            size_t cand = T->t[h]; T->t[h] = (uint32_t)cursor;
            if (cursor != init && cursor - cand <= 0xFFFF) {
                size_t a = cursor, b = cand, lim = len - 5, m = 0;
                while (a + m < lim && in[a + m] == in[b + m]) ++m;
                if (m >= 4) {
                    cursor += m;
                    uint32_t h2 = hash5(in + cursor - 2, len - (cursor - 2));
                    T->t[h2] = (uint32_t)(cursor - 2);
                    break;
                }
            }
            cursor += step; step = sc >> 6; if (ls + 1 != cursor) sc++;
        }
    }
    return len;
}
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); size_t total = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t* data = malloc(total); if (fread(data, 1, total, f) != total) return 1; fclose(f);
    size_t SEG = argc > 2 ? atol(argv[2]) : 262144, W = argc > 3 ? atol(argv[3]) : 131072;
    int nseg = 0, nfail_pos = 0, nfail_tab = 0;
    for (size_t b0 = 0; b0 < total; b0 += BS) {
        size_t n = total - b0 < BS ? total - b0 : BS; const uint8_t* in = data + b0;
        // true parse: snapshots at every boundary
        for (size_t p = SEG; p + 65536 < n; p += SEG) {
            tab_t T; memset(&T, 0, sizeof T); tab_t snapT, snapS; size_t posT = 0, posS = 0; int d;
            parse(in, n, 0, &T, p + 1, p, &snapT, &posT, &d);      // true: from 0 until first seq start >= p (stop right after snapshot)
            tab_t S; memset(&S, 0, sizeof S);
            parse(in, n, p - W, &S, p + 1, p, &snapS, &posS, &d);
            nseg++;
            if (posT != posS) { nfail_pos++; continue; }
            int bad = 0;
            for (int i = 0; i < 4096; ++i) {
                uint32_t a = snapT.t[i], c = snapS.t[i];
                int la = posT - a <= 65535 + 70000 ? 1 : 0;   // live-ish (could still be a candidate later? candidate valid iff cursor - cand <= 0xFFFF at probe time; cursor >= posT)
                la = (posT - (size_t)a) <= 65535; int lc = (posT - (size_t)c) <= 65535;
                if (la != lc || (la && a != c)) { bad = 1; break; }
            }
            if (bad) nfail_tab++;
        }
    }
    printf("SEG %zu W %zu: boundaries %d, start mismatch %d, table mismatch %d -> fail %.1f%%\n", SEG, W, nseg, nfail_pos, nfail_tab, 100.0 * (nfail_pos + nfail_tab) / nseg);
    return 0;
}
