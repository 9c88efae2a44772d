"""Red-zone suite (-m gpu): the kernels stay inside [out, out + out_cap) and do not depend on bytes behind input_len.

The reference cannot write out of bounds by construction (`#![forbid(unsafe_code)]`, src/lib.rs:1) and fuzzes for exactly that
(fuzz/fuzz_targets/decode.rs; overshoot bound src/raw/decompress.rs:55-57).  Here the same property is checked from outside,
tests/redzone.py: every job's output slot sits between 4 KiB zones of poison inside a larger device allocation, its input in front
of poison A in one run and poison B in a second — zones intact, statuses and Ok bytes identical under both poisons, and equal to
the oracle's.  Over: the mutated-block suite, blocks of another encoder (truncated), the segmented pipeline's mixed batch, damaged
4 MiB blocks in batches of 300 and 700, a batch of 3 300 jobs (the bitmap-fed kernel and its hand-over to the pair kernel), the
three forced fall-backs, a time-boxed seeded stress, and compress at out_cap in {0, C - 1, C, N}."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

import oracle_ffi as o
import redzone
import vectors
import rust_lz_fear_amd  # noqa: F401
from rust_lz_fear_amd import ffi, synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
BS = 4 << 20


def _mutated_items(seed=12345):
    from test_gpu_parity import all_cases, mutate
    rng = np.random.default_rng(seed)
    items, exp = [], []
    for name, d in [(n, d) for n, d in all_cases() if 0 < len(d) <= 300000]:
        comp = o.compress2(d)[1]
        for k in range(6):
            m = mutate(rng, comp)
            limit = len(d) if k % 2 == 0 else len(d) // 2 + 1
            cap = limit + len(m) + 64                                # limit + C reproduces every outcome
            items.append(dict(input=m, limit=limit, out_cap=cap)); exp.append(o.decompress_raw(m, limit=limit, cap=cap))
    return items, exp


def test_redzone_mutated_blocks():
    items, exp = _mutated_items()
    res = redzone.check_decompress(items, exp, "mutated blocks")
    assert {0, 1, 2, 3, 4} <= {s for s, _ in res}                    # every DecodeError variant went through the guarded run


def test_redzone_tight_capacity_and_prefix():
    """Output slots with NO slack (out_cap = exactly what the block decodes to, and less), prefix and existing output."""
    d = synth.gen_text_zipf(31, 50000).tobytes()
    dic, payload = d[:20000], d[20000:]
    comp = o.compress2(dic + payload, cursor=len(dic))[1]
    c2 = o.compress2(d)[1]
    items = [dict(input=comp, prefix=dic, limit=len(payload), out_cap=len(payload)),
             dict(input=comp, existing=dic, limit=len(d), out_cap=len(d)),
             dict(input=comp, prefix=dic[:100], limit=len(payload), out_cap=len(payload)),
             dict(input=c2, limit=len(d), out_cap=len(d)),
             dict(input=c2, limit=len(d), out_cap=len(d) - 1),
             dict(input=c2, limit=len(d), out_cap=100),
             dict(input=c2, limit=len(d), out_cap=0),
             dict(input=c2, limit=len(d) - 7, out_cap=len(d))]
    exp = [o.decompress_raw(it["input"], prefix=it.get("prefix", b""), existing=it.get("existing", b""), limit=it["limit"], cap=it["out_cap"]) for it in items]
    redzone.check_decompress(items, exp, "tight capacity")


def test_redzone_blocks_of_another_encoder_truncated():
    J = json.load(open(os.path.join(GOLD, "hc_blocks.json")))
    blob = open(os.path.join(GOLD, "hc_blocks.bin"), "rb").read()
    rng = np.random.default_rng(5)
    items, exp = [], []
    for b in J["blocks"]:
        comp = blob[b["offset"]: b["offset"] + b["length"]]
        data = eval(b["input"], {"synth": synth}).tobytes()
        for cut in (len(comp), len(comp) // 2, int(rng.integers(1, len(comp))), len(comp) - 1):
            m = comp[:cut]
            for limit in (len(data), len(data) - 1):
                cap = limit + len(m) + 64
                items.append(dict(input=m, limit=limit, out_cap=cap)); exp.append(o.decompress_raw(m, limit=limit, cap=cap))
    redzone.check_decompress(items, exp, "HC fixtures")


def test_redzone_segmented_pipeline_mixed_batch():
    from test_gpu_parity import _seg_mixed_items
    items, exp = _seg_mixed_items(np.random.default_rng(77))
    redzone.check_decompress(items, exp, "segmented pipeline, mixed batch")
    assert ffi.lib().lzf_last_decompress_launch().decode().startswith("segmented")
    items, exp = items * 2, exp * 2                                  # 32 jobs and more: the pipeline's last two stages in groups
    redzone.check_decompress(items, exp, "segmented pipeline, grouped")


@pytest.mark.parametrize("n_jobs", [300, 700])
def test_redzone_damaged_4mib_blocks_in_mid_size_batches(n_jobs):
    from test_gpu_hardening import _damage
    rng = np.random.default_rng(2000 + n_jobs)
    raws = [synth.silesia_mix(k * BS, (k + 1) * BS).tobytes() for k in (0, 3, 9, 17, 26, 31, 38, 44)]
    comps = [o.compress2(d)[1] for d in raws]
    bad_at = set(int(x) for x in rng.choice(n_jobs, n_jobs // 10, replace=False))
    items, exp = [], []
    for i in range(n_jobs):
        k = i % len(raws)
        if i in bad_at:
            m = _damage(rng, comps[k], int(rng.integers(0, 4)))
            cap = BS + len(m) + 64
            items.append(dict(input=m, limit=BS, out_cap=cap)); exp.append(o.decompress_raw(m, limit=BS, cap=cap))
        else:
            items.append(dict(input=comps[k], limit=BS, out_cap=BS)); exp.append((0, raws[k]))        # no slack at all behind a good block
    redzone.check_decompress(items, exp, f"{n_jobs} jobs of 4 MiB")
    assert ffi.lib().lzf_last_decompress_launch().decode().startswith("segmented")


def test_redzone_bitmap_fed_kernel_and_its_hand_over():
    """More jobs than the segmented pipeline takes: the bitmap-fed kernel (every job in pieces, drawn by the resident wavefronts) and
    the pair kernel behind it for what it leaves — damaged blocks, prefix, existing output, inputs below its window."""
    from test_gpu_hardening import _damage
    rng = np.random.default_rng(31)
    base = synth.silesia_mix(0, 24 << 20)
    raws, comps = [], []
    for i in range(60):
        a = int(rng.integers(0, (24 << 20) - 300000)); ln = int(rng.choice([3000, 20000, 70000, 150000, 260000, 800000]))
        d = base[a:a + ln].tobytes(); raws.append(d); comps.append(o.compress2(d)[1])
    n_jobs = 3300
    items, exp = [], []
    for i in range(n_jobs):
        k = i % len(raws)
        r = i % 23
        if r == 0:
            m = _damage(rng, comps[k], int(rng.integers(0, 4))); cap = len(raws[k]) + len(m) + 64
            items.append(dict(input=m, limit=len(raws[k]), out_cap=cap)); exp.append(o.decompress_raw(m, limit=len(raws[k]), cap=cap))
        elif r == 1:
            items.append(dict(input=comps[k], limit=len(raws[k]), out_cap=len(raws[k]) - 3)); exp.append(o.decompress_raw(comps[k], limit=len(raws[k]), cap=len(raws[k]) - 3))
        elif r == 2:
            d = raws[k]; cut = len(d) // 3
            cp = o.compress2(d, cursor=cut)[1]
            items.append(dict(input=cp, prefix=d[:cut], limit=len(d) - cut, out_cap=len(d) - cut)); exp.append((0, d[cut:]))
        elif r == 3:
            d = raws[k]; cut = len(d) // 2
            cp = o.compress2(d, cursor=cut)[1]
            items.append(dict(input=cp, existing=d[:cut], limit=len(d), out_cap=len(d))); exp.append((0, d))
        else:
            items.append(dict(input=comps[k], limit=len(raws[k]), out_cap=len(raws[k]))); exp.append((0, raws[k]))
    redzone.check_decompress(items, exp, "3 300 jobs", max_input_len=max(len(it["input"]) for it in items))
    assert ffi.lib().lzf_last_decompress_launch().decode().startswith("bitmap-fed"), ffi.lib().lzf_last_decompress_launch().decode()


@pytest.mark.parametrize("force", ["noscratch", "stager", "resolver"])
def test_redzone_forced_fallbacks(force):
    from rust_lz_fear_amd import build
    env = dict(os.environ, LZF_LIB_PATH=build.build_analysis_library(), LZF_SEG_FORCE=force)
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "redzone_force_check.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "redzone force ok" in r.stdout


def test_redzone_seeded_stress_25s():
    """tests/stress_parity.py's input generator through the guarded entry: compress == oracle with exact-fit and short output slots,
    decompress of the results into exact-fit slots, randomly damaged blocks — for 25 seconds, at least one round."""
    import stress_parity
    rng = np.random.default_rng(606)
    t0 = time.time(); rounds = 0; n_bad = 0
    while rounds < 1 or time.time() - t0 < 25.0:
        data = [stress_parity.make_input(rng, int(rng.choice([200, 5000, 70000, 400000, 1500000]))) for _ in range(16)]
        comps = [o.compress2(d) for d in data]
        citems, cexp = [], []
        for d, (rc, c) in zip(data, comps):
            for cap in sorted({0, max(len(c) - 1, 0), len(c), len(d)}):
                citems.append(dict(input=d, out_cap=cap)); cexp.append(o.compress2(d, cap=cap))
        redzone.check_compress(citems, cexp, f"stress round {rounds}, compress")
        ditems, dexp = [], []
        for d, (rc, c) in zip(data, comps):
            ditems.append(dict(input=c, limit=max(len(d), 1), out_cap=len(d))); dexp.append((0, d))
            if 0 < len(c) <= 400000:
                b = bytearray(c)
                for _ in range(int(rng.integers(1, 4))):
                    b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
                if rng.integers(0, 3) == 0 and len(b) > 2:
                    del b[int(rng.integers(1, len(b))):]
                m = bytes(b); cap = len(d) + len(m) + 64
                ditems.append(dict(input=m, limit=len(d), out_cap=cap)); dexp.append(o.decompress_raw(m, limit=len(d), cap=cap)); n_bad += 1
        redzone.check_decompress(ditems, dexp, f"stress round {rounds}, decompress")
        rounds += 1
    assert n_bad >= 8


def test_redzone_compress_output_caps():
    """compress2 into slots of out_cap in {0, C - 1, C, N} (NoPartialWrites, framed/compress.rs:294-314: a refused write writes nothing
    — here: nothing outside the slot), U32 and U16 tables, cursor > 0, the team kernel (small batch) and the compact kernel (300 jobs)."""
    rng = np.random.default_rng(9)
    datas = [vectors.rng_bytes(11, 5000), synth.gen_text_zipf(3, 70000).tobytes(), synth.silesia_mix(0, 1 << 20).tobytes(),
             synth.gen_exe(5, 200000).tobytes(), bytes(100000), synth.repeat256(65535).tobytes(), b"", b"abc"]
    items, exp = [], []
    for d in datas:
        c = o.compress2(d)[1]
        for cap in sorted({0, max(len(c) - 1, 0), len(c), len(d), len(d) + len(d) // 255 + 16}):
            items.append(dict(input=d, out_cap=cap)); exp.append(o.compress2(d, cap=cap))
        if 0 < len(d) <= 65535:
            c16 = o.compress2(d, kind=o.TABLE_U16)[1]
            for cap in sorted({0, max(len(c16) - 1, 0), len(c16), len(d)}):
                items.append(dict(input=d, kind=ffi.TABLE_U16, out_cap=cap)); exp.append(o.compress2(d, kind=o.TABLE_U16, cap=cap))
        if len(d) > 100:
            cur = len(d) // 3
            cc = o.compress2(d, cursor=cur)[1]
            for cap in sorted({0, max(len(cc) - 1, 0), len(cc)}):
                items.append(dict(input=d, cursor=cur, out_cap=cap)); exp.append(o.compress2(d, cursor=cur, cap=cap))
    redzone.check_compress(items, exp, "compress caps (small batch: the team kernel)")
    # more jobs than compute units: the compact kernel
    base = synth.silesia_mix(0, 2 << 20)
    items, exp = [], []
    for i in range(300):
        a = int(rng.integers(0, (2 << 20) - 3000)); ln = int(rng.integers(1, 2500))
        d = base[a:a + ln].tobytes(); c = o.compress2(d)[1]
        cap = [0, max(len(c) - 1, 0), len(c), len(d)][i % 4]
        items.append(dict(input=d, out_cap=cap)); exp.append(o.compress2(d, cap=cap))
    redzone.check_compress(items, exp, "compress caps (300 jobs: the compact kernel)")
    assert ffi.lib().lzf_last_compress_launch().decode().startswith("lzf_compress_compact_kernel")
