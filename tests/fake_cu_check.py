"""The dispatch thresholds are functions of the device's compute units and LDS (capi.hip `geometry()`), not literals: with the analysis
knob LZF_FAKE_CU=64 the same device dispatches as one with 64 CUs — the segmented pipeline up to 4 x 64 blocks with rings of 128 / 64 /
32 KiB at 64 / 128 / 256 blocks, paired48 up to 8 x 64, paired24 up to 12 x 64, the bitmap-fed kernel beyond (round 6; paired24 /
staged16 behind it when it declines a call) — and every class still decodes to the oracle's bytes.  Run as a script by tests/test_gpu_hardening.py (the knob is read once per process)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np  # noqa: E402
import oracle_ffi as o  # noqa: E402
import rust_lz_fear_amd  # noqa: E402,F401
from rust_lz_fear_amd import ffi, synth  # noqa: E402


def main():
    cu = int(os.environ["LZF_FAKE_CU"])
    rng = np.random.default_rng(8)
    base = synth.silesia_mix(0, 2 << 20)
    big = [synth.silesia_mix(k << 20, (k << 20) + 900000).tobytes() for k in (1, 9, 30, 50)]      # >= 256 KiB compressed: the pipeline's window (64 KiB) and the bitmap-fed kernel's (256 KiB)
    bigc = [o.compress2(d)[1] for d in big]
    assert all(len(c) > 262144 for c in bigc), [len(c) for c in bigc]
    seen = {}
    # more compute units than the pipeline's rank kernels take jobs (one 1024-thread workgroup, capi.hip kSegRankMax): with
    # LZF_FAKE_CU=512 the pipeline's limit is 1024 jobs, not 4 x 512 — 1000 jobs go through it, 1500 go to the pair kernel
    sizes = (1000, 1500) if cu > 256 else (cu - 4, cu + 4, 2 * cu + 4, 4 * cu - 4, 4 * cu + 8, 8 * cu + 8, 13 * cu + 8, 64 * cu + 16)
    for n in sizes:
        raws, comps = [], []
        for i in range(n):
            if i % 16 == 0:
                k = (i // 16) % len(big); raws.append(big[k]); comps.append(bigc[k])
            else:
                a = int(rng.integers(0, (2 << 20) - 3000)); ln = int(rng.integers(1, 2500))
                d = base[a:a + ln].tobytes(); raws.append(d); comps.append(o.compress2(d)[1])
        res = ffi.decompress_blocks_host([dict(input=c, limit=max(len(d), 1), out_cap=len(d) + len(c) + 64) for d, c in zip(raws, comps)])
        launch = ffi.lib().lzf_last_decompress_launch().decode()
        for i, (d, (rc, out)) in enumerate(zip(raws, res)):
            assert rc == 0 and out == d, (n, i, rc)
        seen[n] = launch
        print(n, launch, flush=True)
    if cu > 256:
        assert seen[1000].startswith("segmented"), seen[1000]
        assert not seen[1500].startswith("segmented") and "lzf_decompress_paired_kernel<4096,48,640>" in seen[1500], seen[1500]
        print("geometry ok")
        return
    want = {cu - 4: "resolve_pair_kernel<131072>", cu + 4: "resolve_pair_kernel<65536>", 2 * cu + 4: "resolve_pair_kernel<32768>", 4 * cu - 4: "resolve_pair_kernel<32768>",
            4 * cu + 8: "lzf_decompress_paired_kernel<4096,48,640>", 8 * cu + 8: "lzf_decompress_paired_kernel<4096,24,384>", 13 * cu + 8: "bitmap-fed", 64 * cu + 16: "bitmap-fed"}
    for n, frag in want.items():
        assert frag in seen[n], (n, seen[n], frag)
        if "resolve" not in frag:
            assert not seen[n].startswith("segmented"), (n, seen[n])
    print("geometry ok")


if __name__ == "__main__":
    main()
