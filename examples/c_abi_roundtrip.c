/* c_abi_roundtrip.c — the drop-in boundary used from plain C (no Python, no C++):
 *   gcc examples/c_abi_roundtrip.c -Iinclude -Lrust-lz-fear_amd -llzfear_hip -Wl,-rpath,$PWD/rust-lz-fear_amd -o c_abi_roundtrip
 * One compress2 job and one decompress_raw job through the host-buffer entry points, then a
 * default-settings frame through the frame layer.  Exit code 0 = round trips are exact. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "lzfear_hip.h"
#include "lzfear_frame.h"

int main(void) {
    const size_t n = 3u << 20;
    unsigned char* data = malloc(n);
    unsigned long long s = 42;
    static const char* words[16] = { "block ", "codec ", "frame ", "window ", "offset ", "literal ", "match ", "token ",
                                     "table ", "cursor ", "hash ", "wave ", "lane ", "ring ", "chunk ", "stream\n" };
    for (size_t i = 0; i < n;) {                     /* compressible: a stream of words */
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const char* w = words[(s >> 60) & 15];
        for (size_t k = 0; w[k] && i < n; ++k) data[i++] = (unsigned char)w[k];
    }
    if (lzf_abi_version() != LZFEAR_ABI_VERSION) return 10;
    if (lzf_device_count() < 1) { fprintf(stderr, "%s\n", lzf_last_error()); return 11; }

    /* raw::compress2(input, 0, &mut U32Table::default(), NoPartialWrites(cap = n)) */
    unsigned char* comp = malloc(n);
    lzf_compress_job cj = { data, n, 0, comp, n, NULL, LZF_TABLE_U32, 0 };
    lzf_job_result cr;
    if (lzf_compress_batch_host(&cj, &cr, 1) != LZF_OK || cr.status != LZF_OK) return 12;

    /* raw::decompress_raw(input, &[], &mut Vec::new(), n) */
    unsigned char* back = malloc(n + cr.out_len);
    lzf_decompress_job dj = { comp, cr.out_len, NULL, 0, back, 0, n + cr.out_len, n };
    lzf_job_result dr;
    if (lzf_decompress_batch_host(&dj, &dr, 1) != LZF_OK || dr.status != LZF_OK) return 13;
    if (dr.out_len != n || memcmp(back, data, n) != 0) return 14;

    /* CompressionSettings::default().compress(..) / decompress_frame(..) */
    lzf_settings st;
    lzf_settings_default(&st);
    st.block_size = 1u << 20;
    size_t cap = lzf_frame_compress_bound(&st, n), flen = 0, olen = 0, used = 0;
    unsigned char* frame = malloc(cap);
    if (lzf_frame_compress(&st, data, n, frame, cap, &flen) != LZF_OK) return 15;
    if (lzf_frame_decompress(frame, flen, NULL, 0, back, n + cr.out_len, &olen, &used) != LZF_OK) return 16;
    if (olen != n || used != flen || memcmp(back, data, n) != 0) return 17;

    /* a malformed block reports the reference's error kind, never crashes */
    unsigned char bad[4] = { 0x10, 'a', 2, 0 };      /* src/raw/decompress.rs:173 offset_oob */
    lzf_decompress_job bj = { bad, 4, NULL, 0, back, 0, 64, 64 };
    if (lzf_decompress_batch_host(&bj, &dr, 1) != LZF_OK || dr.status != LZF_INVALID_DEDUP_OFFSET) return 18;

    printf("c abi ok: %zu -> %llu bytes (block), %zu bytes (frame)\n", n, (unsigned long long)cr.out_len, flen);
    return 0;
}
