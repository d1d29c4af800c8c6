#!/usr/bin/env python3
"""Micro-benchmark of the two attention kernels alone (for A/B work and PMC passes).

    python tools/bench_kernels.py prefill --nf 4 --past 65536 --chunk 16384 --reps 5
    python tools/bench_kernels.py decode  --ctx 131072 --reps 20        # all 32 layers of the bench pattern
Prints one JSON line per case: avg ms, algorithmic TFLOP/s or GB/s.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "duo-attention_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

HQ, HKV, D, W = 32, 8, 128, 384


def pools(n_heads, rows, dev, g):
    k = torch.randn(n_heads, rows, D, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16).permute(1, 0, 2)
    v = torch.randn(n_heads, rows, D, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16).permute(1, 0, 2)
    return k, v


def time_it(fn, reps, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(x.elapsed_time(y) for x, y in evs)
    return sum(ts) / len(ts), ts[0], ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["prefill", "decode", "decode_int4", "linear"])
    ap.add_argument("--rows", type=int, default=1, help="linear: token rows (batch size of the decode step)")
    ap.add_argument("--nf", type=int, default=4)
    ap.add_argument("--past", type=int, default=65536)
    ap.add_argument("--chunk", type=int, default=16384)
    ap.add_argument("--ctx", type=int, default=131072)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--uniform-nf", type=int, default=-1, help="decode: every layer has this many retrieval kv heads")
    a = ap.parse_args()
    from duo_attn import _hip
    from duo_attn.backend import get_backend

    be = get_backend()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    G = HQ // HKV
    if a.what == "linear":
        # the four token-row linears of a Llama-3-8B decoder layer's decode step (csrc/duo_linear.hip) against the same
        # modules through torch (hipBLASLt GEMM at M = rows + separate norm / activation / add kernels); 32 distinct
        # weight sets so that nothing is served from L2 / MALL
        import torch.nn.functional as F
        Hd, I, L = 4096, 14336, 32
        rn = lambda *s_: (torch.randn(*s_, generator=g, device=dev, dtype=torch.float32) * 0.02).to(torch.bfloat16)
        Ws = [dict(q=rn(4096, Hd), k=rn(1024, Hd), v=rn(1024, Hd), o=rn(Hd, 4096), g=rn(I, Hd), u=rn(I, Hd), d=rn(Hd, I),
                   n1=rn(Hd).abs() + 1, n2=rn(Hd).abs() + 1) for _ in range(L)]
        x = rn(a.rows, Hd) * 50
        ao = rn(a.rows, 4096) * 50
        cases = {
            "qkv (rmsnorm prologue)": (lambda w: be.token_linear(x, [(w["q"], None), (w["k"], None), (w["v"], None)], norm=(w["n1"], 1e-5)),
                                       lambda w: [F.linear(be.rmsnorm(x, w["n1"], 1e-5), w[n]) for n in "qkv"], 6144 * Hd),
            "o_proj + residual": (lambda w: be.token_linear(ao, [(w["o"], None)], residual=x),
                                  lambda w: x + F.linear(ao, w["o"]), Hd * 4096),
            "gate|up (rmsnorm prologue)": (lambda w: be.token_linear(x, [(w["g"], None), (w["u"], None)], norm=(w["n2"], 1e-5)),
                                           lambda w: [F.linear(be.rmsnorm(x, w["n2"], 1e-5), w[n]) for n in "gu"], 2 * I * Hd),
        }
        gu = be.token_linear(x, [(Ws[0]["g"], None), (Ws[0]["u"], None)], norm=(Ws[0]["n2"], 1e-5))
        cases["down_proj (silu*mul prologue) + residual"] = (
            lambda w: be.token_linear(gu[:, :I], [(w["d"], None)], x2=gu[:, I:], residual=x),
            lambda w: x + F.linear(F.silu(gu[:, :I]) * gu[:, I:], w["d"]), Hd * I)
        tot = {"hip": 0.0, "torch": 0.0}
        for name, (hip_fn, torch_fn, nelem) in cases.items():
            res = {}
            for tag, fn in (("hip", hip_fn), ("torch", torch_fn)):
                avg, mn, med = time_it(lambda: [fn(w) for w in Ws], a.reps)
                res[tag] = avg / L * 1e3
                tot[tag] += avg / L * 1e3
            print(json.dumps({"case": f"{name} rows={a.rows}", "weight_MB": nelem * 2 / 1e6, "hip_us": round(res["hip"], 2),
                              "torch_us": round(res["torch"], 2), "hip_TBps": round(nelem * 2 / res["hip"] / 1e6, 3),
                              "torch_TBps": round(nelem * 2 / res["torch"] / 1e6, 3)}))
        wbytes = (6144 * Hd + Hd * 4096 + 3 * I * Hd) * 2
        print(json.dumps({"case": f"layer total rows={a.rows}", "hip_us": round(tot["hip"], 1), "torch_us": round(tot["torch"], 1),
                          "hip_TBps": round(wbytes / tot["hip"] / 1e6, 3), "torch_TBps": round(wbytes / tot["torch"] / 1e6, 3),
                          "x32_layers_ms": {k: round(v * 32 / 1e3, 3) for k, v in tot.items()}}))
        return
    scale = D ** -0.5
    if a.what == "prefill":
        nf, ns, S, past = a.nf, HKV - a.nf, a.chunk, a.past
        q = torch.randn(S, HQ, D, generator=g, device=dev).to(torch.bfloat16)
        kn, vn = pools(HKV, S, dev, g)
        out = torch.empty_like(q)
        fk, fv = pools(max(nf, 1), past + S, dev, g)
        sk, sv = pools(max(ns, 1), W, dev, g)
        full = (nf, 0, (fk[:past, :nf], fv[:past, :nf]), (fk[past:past + S, :nf], fv[past:past + S, :nf])) if nf else None
        stream = (ns, nf * G, (sk[:, :ns], sv[:, :ns]), (kn[:, nf:], vn[:, nf:])) if ns else None
        _hip.set_debug_flags(a.flags)
        avg, mn, med = time_it(lambda: be.attention(q, out, G, full, stream, scale), a.reps)
        tri = S * (S + 1) / 2
        flops = 4 * D * G * (nf * (S * past + tri) + ns * (S * min(past, W) + tri))
        print(json.dumps({"case": f"prefill nf={nf} past={past} S={S}", "avg_ms": avg, "min_ms": mn,
                          "tflops_avg": flops / avg / 1e9, "tflops_best": flops / mn / 1e9}))
    elif a.what == "decode_int4":
        import bench

        counts = bench.LLAMA3_8B_FULL_KV_HEADS
        N = a.ctx
        q = torch.randn(HQ, D, generator=g, device=dev).to(torch.float16)
        out = torch.empty_like(q)
        layers = []
        for nf in counts:
            ns = HKV - nf
            mk = lambda h, T: (torch.randint(0, 256, (h, T, 64), generator=g, device=dev, dtype=torch.uint8).permute(1, 0, 2),
                               (torch.rand(h, T, 2, generator=g, device=dev) * 0.3 + 0.01).to(torch.float16).permute(1, 0, 2))
            fkq, fksz = mk(max(nf, 1), N + 1)
            fvq, fvsz = mk(max(nf, 1), N + 1)
            skq, sksz = mk(max(ns, 1), W + 1)
            svq, svsz = mk(max(ns, 1), W + 1)
            full = _hip.make_int4_pool(fkq[:, :nf], fksz[:, :nf], fvq[:, :nf], fvsz[:, :nf], N + 1, 0) if nf else None
            stream = _hip.make_int4_pool(skq[:, :ns], sksz[:, :ns], svq[:, :ns], svsz[:, :ns], W + 1, nf * G) if ns else None
            layers.append((full, stream, (fkq, fksz, fvq, fvsz, skq, sksz, svq, svsz)))

        _hip.set_debug_flags(a.flags)

        def step():
            for full, stream, _ in layers:
                _hip.attn_decode_int4(q, out, G, full, stream, scale)

        avg, mn, med = time_it(step, a.reps)
        rows = sum(nf * (N + 1) + (HKV - nf) * (W + 1) for nf in counts)
        nbytes = rows * 2 * 68
        print(json.dumps({"case": f"int4 decode (split+merge) x32 layers ctx={N}", "avg_ms": avg, "min_ms": mn,
                          "GBps_avg": nbytes / avg / 1e6, "rows_per_us": rows / avg / 1e3,
                          "bf16_equivalent_GBps": rows * 512 / avg / 1e6}))
    else:
        import bench

        counts = bench.LLAMA3_8B_FULL_KV_HEADS if a.uniform_nf < 0 else [a.uniform_nf] * 8
        N = a.ctx
        q = torch.randn(1, HQ, D, generator=g, device=dev).to(torch.bfloat16)
        kn, vn = pools(HKV, 1, dev, g)
        out = torch.empty_like(q)
        layers = []
        for nf in counts:
            ns = HKV - nf
            fk, fv = pools(max(nf, 1), N + 1, dev, g)
            sk, sv = pools(max(ns, 1), W, dev, g)
            full = (nf, 0, (fk[:N, :nf], fv[:N, :nf]), (fk[N:N + 1, :nf], fv[N:N + 1, :nf])) if nf else None
            stream = (ns, nf * G, (sk[:, :ns], sv[:, :ns]), (kn[:, nf:], vn[:, nf:])) if ns else None
            layers.append((full, stream))
        _hip.set_debug_flags(a.flags | 2)

        def step():
            for full, stream in layers:
                be.attention(q, out, G, full, stream, scale)

        avg, mn, med = time_it(step, a.reps)
        nbytes = sum(bench.decode_bytes(counts, N))
        print(json.dumps({"case": f"decode split kernel x{len(counts)} layers ctx={N}" + (f" nf={a.uniform_nf}" if a.uniform_nf >= 0 else ""), "avg_ms": avg, "min_ms": mn,
                          "GBps_avg": nbytes / avg / 1e6, "GBps_best": nbytes / mn / 1e6,
                          "frac_of_8TBps": nbytes / avg / 1e6 / 8000}))
        _hip.set_debug_flags(0)


if __name__ == "__main__":
    main()
