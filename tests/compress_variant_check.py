"""Parity check of whatever compress kernel LZF_COMPRESS_KERNEL selects ("general" keeps fresh-table U32 jobs on
lzf_compress_wave_kernel, "compact" switches the latency class off; default for a batch of this size = lzf_compress_team_kernel).  Run as a script by
tests/test_gpu_parity.py::test_every_compress_kernel (the choice is read once per process, hence the subprocess)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np  # noqa: E402
import oracle_ffi as o  # noqa: E402
import vectors  # noqa: E402
import rust_lz_fear_amd  # noqa: E402,F401
from rust_lz_fear_amd import ffi, synth  # noqa: E402


def main():
    cases = vectors.small_cases() + vectors.medium_cases()
    # positions beyond several 64 KiB epochs, long matches that skip epochs, incompressible stretches (wide batches)
    rng = np.random.default_rng(5)
    big = synth.silesia_mix(30 << 20, (30 << 20) + 700_000).tobytes()
    zeros = bytes(300_000)
    noise = rng.integers(0, 256, 200_000, dtype=np.uint8).tobytes()
    cases += [("mix700k", big), ("zeros_noise_zeros", zeros + noise + zeros + big[:70_000] + zeros[:140_000] + big[:70_000]),
              ("noise_then_repeat", noise + noise[:150_000] + big[:100_000])]
    res = ffi.compress_blocks_host([dict(input=d, out_cap=len(d) + len(d) // 200 + 64) for _, d in cases])
    launch = ffi.lib().lzf_last_compress_launch().decode()
    want = {"general": "lzf_compress_wave_kernel", "compact": "lzf_compress_compact_kernel"}.get(os.environ.get("LZF_COMPRESS_KERNEL", ""), "lzf_compress_team_kernel")
    assert launch.startswith(want), (launch, want)
    for (name, d), (rc, out) in zip(cases, res):
        erc, eout = o.compress2(d)
        assert rc == erc and out == eout, name
    # cursor > 0 with a fresh table (prefix as history) and capacity edges
    items, exp = [], []
    for name, d in cases[-3:] + vectors.medium_cases()[:2]:
        for cur in (1, 4096, 65536, 140_000):
            if cur >= len(d):
                continue
            for cap in (len(d), 1000):
                items.append(dict(input=d, cursor=cur, out_cap=cap)); exp.append(o.compress2(d, cursor=cur, cap=cap))
    res = ffi.compress_blocks_host(items)
    for it, (rc, out), (erc, eout) in zip(items, res, exp):
        assert rc == erc and (rc != 0 or out == eout), (len(it["input"]), it["cursor"], it["out_cap"])
    print("variant ok")


if __name__ == "__main__":
    main()
