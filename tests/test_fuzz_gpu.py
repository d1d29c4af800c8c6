"""The randomised differential run of the static hot path (tests/fuzz_static_path.py) as tests.

* CPU: the harness itself — a handful of drawn cases with the oracle plugged in as the device backend (oracle against
  oracle through the product's host plumbing: batch rows, row blocks, evictions), so a harness error cannot hide on the GPU;
* GPU: the cases the fuzzer FOUND (kept as regressions), and a fixed seed's first cases.

What it found (round 4): a batched (B > 1) fused decode step with the retrieval heads split over several workgroups and the
streaming-pool update folded into the scan stored every batch row's partials into row 0's workspace area — rows >= 1 merged
unwritten partials.  No earlier B > 1 decode test had a context long enough to split a retrieval head."""
import random

import pytest

import fuzz_static_path as F

FOUND = [
    # B = 2 decode, retrieval heads split, saturated streaming pool (sink + recent < context), group 1 / 2 / 4
    dict(Hkv=2, group=1, counts=[1, 1], sink=2, recent=300, chunks=[30, 526, 679], row_block=None, decode_steps=2, evict=False,
         theta=10000.0, rope_scale=1.0, scale=1.0, B=2, seed=1233713562),
    dict(Hkv=8, group=4, counts=[8, 5], sink=2, recent=256, chunks=[588], row_block=None, decode_steps=2, evict=True,
         theta=1000000.0, rope_scale=4.0, scale=0.5, B=2, seed=412307310),
    dict(Hkv=4, group=2, counts=[1, 4], sink=64, recent=100, chunks=[590], row_block=None, decode_steps=1, evict=False,
         theta=1000000.0, rope_scale=1.0, scale=1.0, B=2, seed=338587798),
    dict(Hkv=8, group=2, counts=[0, 4], sink=128, recent=3, chunks=[53, 593], row_block=None, decode_steps=4, evict=True,
         theta=1000000.0, rope_scale=4.0, scale=1.0, B=2, seed=1156546539),
    dict(Hkv=3, group=2, counts=[0, 2], sink=64, recent=3, chunks=[126, 114, 208, 1], row_block=None, decode_steps=1, evict=False,
         theta=500000.0, rope_scale=1.0, scale=0.5, B=2, seed=1762049565),
    # large-magnitude data (peaky softmax), odd group sizes, row blocks
    dict(Hkv=3, group=3, counts=[0], sink=1, recent=256, chunks=[328, 528], row_block=None, decode_steps=4, evict=True,
         theta=10000.0, rope_scale=1.0, scale=2.5, B=1, seed=1045457843),
    dict(Hkv=8, group=4, counts=[8], sink=16, recent=8, chunks=[40, 1575, 480, 683], row_block=512, decode_steps=4, evict=False,
         theta=10000.0, rope_scale=1.0, scale=2.5, B=1, seed=1093887530),
    dict(Hkv=2, group=7, counts=[1, 2], sink=4, recent=32, chunks=[17, 300, 1], row_block=64, decode_steps=3, evict=True,
         theta=500000.0, rope_scale=1.0, scale=1.0, B=2, seed=5),
    # round 5: low-variance data (x 0.5: a flat softmax whose weights straddle 0.5, where one bf16 rounding is relatively
    # largest) — the reference's OWN arithmetic (bf16 P, bf16 output) is 2.550e-3 of rms away from exact P on the second
    # chunk's streaming-only layer, the HIP kernel 2.549e-3: past the 2.5e-3 bar, not past the reference
    # (fuzz_static_path._no_noisier_than_the_reference_arithmetic)
    dict(Hkv=1, group=8, counts=[1, 0], sink=64, recent=256, chunks=[1344, 551], row_block=None, decode_steps=4, evict=True,
         graph=True, theta=500000.0, rope_scale=1.0, scale=0.5, B=1, seed=483580883),
]


def test_fuzz_harness_on_the_cpu_with_the_oracle_backend(monkeypatch):
    from duo_attn import backend
    from oracle.duo_oracle import OracleBackend

    monkeypatch.setattr(F, "DEV", "cpu")
    backend._set_backend_for_testing(OracleBackend(round_p=False))      # (the expectation is the exact-P oracle)
    try:
        rng = random.Random(4)
        n = 0
        while n < 12:
            c = F.draw_case(rng)
            if sum(c["chunks"]) > 900:
                continue
            F.run_case(c)
            n += 1
    finally:
        backend._set_backend_for_testing(None)


@pytest.mark.gpu
@pytest.mark.parametrize("case", FOUND, ids=lambda c: f"kv{c['Hkv']}g{c['group']}B{c['B']}n{sum(c['chunks'])}")
def test_cases_the_fuzzer_found(case):
    F.run_case(case)


@pytest.mark.gpu
def test_fixed_seed_prefix():
    rng = random.Random(11)
    for _ in range(25):
        F.run_case(F.draw_case(rng))


@pytest.mark.gpu
def test_int4_decode_fixed_seed_prefix():
    """tests/fuzz_int4_decode.py, default (dequantising) kernel: 60 drawn shapes — any group size up to 16, 1 ... 60 000 rows,
    head-major and token-major pools, rows with extreme scales"""
    import fuzz_int4_decode as I

    rng = random.Random(21)
    for _ in range(60):
        I.run_case(I.draw_case(rng))


@pytest.mark.gpu
def test_tuple_path_fixed_seed_prefix():
    """tests/fuzz_tuple_path.py: the tuple-cache attention forward (enable_duo_attention_eval's) on 40 drawn geometries /
    step sequences — outputs at the attention bar, returned caches bit for bit"""
    import fuzz_tuple_path as T

    rng = random.Random(31)
    for _ in range(40):
        T.run_case(T.draw_case(rng))


@pytest.mark.gpu
def test_token_linear_fixed_seed_prefix():
    """tests/fuzz_token_linear.py: 150 drawn shapes of the token-row linears (rows 1-4, any feature count that fits, 1-3
    blocks, padded rows, bias / residual, all four prologues)"""
    import fuzz_token_linear as L

    rng = random.Random(41)
    for _ in range(150):
        L.run_case(L.draw_case(rng))


@pytest.mark.gpu
def test_model_level_forms_agree_fixed_seed_prefix():
    """tests/fuzz_model_decode.py: 40 drawn tiny HF models (Llama / Mistral, random head geometry, MLP width, biases, batch
    rows, patterns) — static path module by module == fused == the reference loop under the automatic HIP graph; tuple path
    module by module == fused"""
    import fuzz_model_decode as M

    rng = random.Random(51)
    for _ in range(40):
        M.run_case(M.draw_case(rng))


@pytest.mark.gpu
def test_int4_cache_flows_fixed_seed_prefix_and_one_token_first_chunk():
    """tests/fuzz_int4_cache.py: put / chunked-prefill attention / fused INT4 decode / compress flows of
    DuoAttentionStaticINT4KVCache on drawn geometries; first the case the fuzzer found — a ONE-TOKEN first chunk (a one-token
    prompt) with two batch rows, which the fp16 dispatch used to refuse"""
    import fuzz_int4_cache as C

    C.run_case(dict(Hkv=1, group=4, counts=[1], sink=16, recent=8, chunk=200, chunks=[1, 57], decode_steps=4, B=2, scale=0.5,
                    seed=667356813))
    C.run_case(dict(Hkv=2, group=2, counts=[1, 0], sink=4, recent=8, chunk=64, chunks=[1, 1, 30], decode_steps=2, B=1, scale=1.0,
                    seed=3))
    rng = random.Random(61)
    for _ in range(40):
        C.run_case(C.draw_case(rng))
