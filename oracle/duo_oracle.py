"""CPU oracle for the DuoAttention split-head attention hot path.

TEST INFRASTRUCTURE ONLY.  Imported by ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` — never by the product package
(``duo-attention_amd/duo_attn``), which has no CPU path.

What it restates (torch, CPU, fp32 math on the same bf16 inputs):

  control flow / cache   reference duo_attn/patch/llama.py:309-434 (static forward),
                         :146-306 (tuple forward), duo_attn/patch/static_kv_cache.py:109-167,
                         252-263, 285-297 (put / compress / split / clear / evict)
  attention arithmetic   flash-attn==2.6.3 ``flash_attn_func(q,k,v,causal=True)`` (un-vendored
                         dependency, README.md:44): softmax(q k^T / sqrt(D) + mask) v, fp32
                         accumulate, GQA h -> h // G, causal mask bottom-right aligned
                         (query i sees keys j <= i + Sk - Sq), P rounded to the input dtype
                         before P.V
  RoPE                   flashinfer ``rope.apply_rope_inplace(interleave=False)`` (un-vendored,
                         README.md:49) as called from duo_attn/patch/flashinfer_utils.py:48-56:
                         rotate-half pairs (i, i+D/2), angle = pos / rope_scale * theta^(-2i/D),
                         fp32
  RMSNorm                flashinfer ``norm.rmsnorm``: x * rsqrt(mean(x^2) + eps) * w in fp32
  decoder layer at       the module sequence either side of the attention op — norms, q/k/v_proj, o_proj, SwiGLU MLP,
  q_len == 1             residual adds (duo_attn/patch/static_kv_cache.py:507-546 around llama.py:309-434) — as
                         ``token_linear_ref``: fp64 dot products, every intermediate the modules materialise rounded to
                         the model dtype; pinned by tests/golden/layer_a.npz (the reference's own layer forward)
  tuple-cache decode     ``tuple_decode_prep_ref``: HF rotary in torch's bf16 arithmetic (transformers
  step, data movement    apply_rotary_pos_emb as called at llama.py:177-184), the cache ++ new row concatenations and the
                         sink/recent truncation (llama.py:202-223, :273-301) written with the reference's own torch ops;
                         ``rmsnorm_hf_ref``: HuggingFace's LlamaRMSNorm.forward (the tuple path keeps HF's norms)

Pinning: the reference has no tests or golden vectors.  The oracle is
pinned against outputs of the REFERENCE'S OWN CODE run in the build container
with the absent third-party packages stubbed by independent restatements of their
published semantics — see tests/golden/make_golden.py and
tests/test_oracle_golden.py (five frozen runs), and tests/golden/fuzz_against_reference.py: the
same construction LIVE over drawn cases — the reference's static forward + cache, tuple forward,
decoder-layer forward, whole models through its enablers, INT4 demo cache and host utilities next to
this oracle and to the product's host path, 71 343 cases with no difference beyond one bf16 ulp (profiles/r4_oracle_vs_reference_fuzz.txt; a
20-second slice runs as a CPU test where /root/reference exists).  The attention/RoPE arithmetic itself lives in
packages that are not under /root/reference, so for that part parity is anchored
on their documented semantics and cross-checked against
torch.nn.functional.scaled_dot_product_attention with an explicit mask.
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import numpy as np
import torch


# ----------------------------------------------------------------------------- RoPE
def rope_inv_freq(rope_theta: float, rope_scale: float, head_dim: int) -> torch.Tensor:
    """theta^(-2i/D) / scale in float64, rounded once to float32 (the product's kernel does the
    same on the host, so the fp32 angle pos * inv_freq is reproducible bit for bit)."""
    theta = float(np.float32(rope_theta))
    scale = float(np.float32(rope_scale))
    i = np.arange(head_dim // 2, dtype=np.float64)
    inv = np.power(theta, -2.0 * i / head_dim) / scale
    return torch.from_numpy(inv.astype(np.float32))


def rope_ref(x: torch.Tensor, pos0: int, rope_scale: float, rope_theta: float) -> torch.Tensor:
    """x: [S, H, D] -> rotated copy in x.dtype (fp32 math)."""
    S, H, D = x.shape
    inv = rope_inv_freq(rope_theta, rope_scale, D)
    pos = torch.arange(pos0, pos0 + S, dtype=torch.float32)
    ang = pos[:, None] * inv[None, :]                       # fp32 multiply
    cos, sin = torch.cos(ang)[:, None, :], torch.sin(ang)[:, None, :]
    xf = x.float()
    lo, hi = xf[..., : D // 2], xf[..., D // 2:]
    out = torch.cat([lo * cos - hi * sin, hi * cos + lo * sin], dim=-1)
    return out.to(x.dtype)


def apply_rope_inplace_ref(q: torch.Tensor, k: torch.Tensor, offsets, rope_scale: float, rope_theta: float):
    """q [B,S,Hq,D], k [B,S,Hkv,D], in place (reference flashinfer_utils.py:29-59)."""
    bsz = q.shape[0]
    if isinstance(offsets, torch.Tensor):
        offs = (offsets.expand(bsz) if offsets.numel() == 1 else offsets).tolist()
    elif isinstance(offsets, (list, tuple)):
        offs = list(offsets)
    else:
        offs = [int(offsets)] * bsz
    for b in range(bsz):
        q[b].copy_(rope_ref(q[b], int(offs[b]), rope_scale, rope_theta))
        k[b].copy_(rope_ref(k[b], int(offs[b]), rope_scale, rope_theta))
    return q, k


# ----------------------------------------------------------------------------- attention
def flash_attn_func_ref(q, k, v, causal=True, dropout_p=0.0, softmax_scale=None, round_p=True,
                        out_dtype=None, return_budget=False):
    """Dense restatement of flash_attn_func.  q [B,Sq,Hq,D]; k,v [B,Sk,Hkv,D].

    ``return_budget``: also return A = softmax(S) . |V| (fp32, same shape as the output).  An
    implementation that rounds P to bf16 before P.V (FA2, and this repo's MFMA kernel) may differ from
    the exact result by at most 2^-9 * A per element (each p_j carries a relative error <= 2^-9)."""
    B, Sq, Hq, D = q.shape
    Sk, Hkv = k.shape[1], k.shape[2]
    G = Hq // Hkv
    scale = softmax_scale if softmax_scale is not None else 1.0 / math.sqrt(D)
    qf = q.float().permute(0, 2, 1, 3)                                  # [B,Hq,Sq,D]
    kf = k.float().permute(0, 2, 1, 3).repeat_interleave(G, dim=1)      # [B,Hq,Sk,D]
    vf = v.float().permute(0, 2, 1, 3).repeat_interleave(G, dim=1)
    s = torch.matmul(qf, kf.transpose(-1, -2)) * scale
    if causal:
        i = torch.arange(Sq)[:, None]
        j = torch.arange(Sk)[None, :]
        s = s.masked_fill(~(j <= i + (Sk - Sq)), float("-inf"))
    if round_p:
        # FA2 normalises AFTER the P.V product: O = (sum_j bf16(exp(s_j - m)) v_j) / sum_j exp(s_j - m)
        m = s.amax(dim=-1, keepdim=True)
        e = torch.exp(s - m)
        l = e.sum(dim=-1, keepdim=True)
        o = torch.matmul(e.to(q.dtype).float(), vf) / l
    else:
        o = torch.matmul(torch.softmax(s, dim=-1), vf)
    o = o.permute(0, 2, 1, 3).to(out_dtype or q.dtype)
    if return_budget:
        a = torch.matmul(torch.softmax(s, dim=-1), vf.abs()).permute(0, 2, 1, 3)
        return o, a
    return o


def duo_visible_mask(kind: str, N: int, S: int, sink: int, recent: int) -> torch.Tensor:
    """[S, N+S] boolean visibility from the closed-form semantics of SURVEY §8(a7) — an independent
    statement of the same thing ``static_forward_ref`` computes procedurally.  Query p = N + i:
      retrieval head: keys {0..p};
      streaming head: N == 0 -> {0..p}; else Pool(N) ∪ {N..p}, Pool(t) = {0..t-1} if t <= W else
      {0..sink-1} ∪ {t-recent..t-1}."""
    W = sink + recent
    j = torch.arange(N + S)[None, :]
    p = (N + torch.arange(S))[:, None]
    causal = j <= p
    if kind == "full" or N == 0:
        return causal
    pool = (j < N) if N <= W else ((j < sink) | ((j >= N - recent) & (j < N)))
    return causal & (pool | (j >= N))


# ----------------------------------------------------------------------------- static cache
class StaticCacheRef:
    """Plain restatement of DuoAttentionStaticKVCache (reference static_kv_cache.py:18-315) with the
    reference's token-major [B, T, h, D] tensors."""

    def __init__(self, num_layers, num_kv_heads, head_dim, full_attention_heads, batch_size, max_size,
                 sink_size, recent_size, dtype=torch.bfloat16):
        self.batch_size, self.max_size = batch_size, max_size
        self.sink_size, self.recent_size = sink_size, recent_size
        self.num_layers, self.num_kv_heads, self.head_dim = num_layers, num_kv_heads, head_dim
        self.num_full_kv_head_list, self.num_streaming_kv_head_list = [], []
        self.kv_seq_len_list = [0] * num_layers
        self.streaming_kv_seq_len_list = [0] * num_layers
        self.full_key_states_list, self.full_value_states_list = [], []
        self.streaming_key_states_list, self.streaming_value_states_list = [], []
        W = sink_size + recent_size
        for heads in full_attention_heads:
            nf = int((torch.as_tensor(heads) > 0.5).sum().item())
            ns = num_kv_heads - nf
            self.num_full_kv_head_list.append(nf)
            self.num_streaming_kv_head_list.append(ns)
            self.full_key_states_list.append(torch.zeros(batch_size, max_size, nf, head_dim, dtype=dtype))
            self.full_value_states_list.append(torch.zeros(batch_size, max_size, nf, head_dim, dtype=dtype))
            self.streaming_key_states_list.append(torch.zeros(batch_size, W, ns, head_dim, dtype=dtype))
            self.streaming_value_states_list.append(torch.zeros(batch_size, W, ns, head_dim, dtype=dtype))

    @property
    def kv_seq_len(self):
        return self.kv_seq_len_list[-1]

    @property
    def streaming_kv_seq_len(self):
        return self.streaming_kv_seq_len_list[-1]

    def split_kv(self, l, k, v):
        nf = self.num_full_kv_head_list[l]
        return k[:, :, :nf], v[:, :, :nf], k[:, :, nf:], v[:, :, nf:]

    def put_full_kv(self, l, fk, fv):
        n, cur = fk.shape[1], self.kv_seq_len_list[l]
        if n + cur > self.max_size:
            raise ValueError(
                f"Trying to put {n} KVs into a cache with max size {self.max_size}, current size: {cur}."
            )
        self.full_key_states_list[l][:, cur:cur + n].copy_(fk)
        self.full_value_states_list[l][:, cur:cur + n].copy_(fv)
        self.kv_seq_len_list[l] += n
        return self.get_full_kv(l)

    def get_full_kv(self, l):
        n = self.kv_seq_len_list[l]
        return self.full_key_states_list[l][:, :n], self.full_value_states_list[l][:, :n]

    def get_streaming_kv(self, l):
        n = self.streaming_kv_seq_len_list[l]
        return self.streaming_key_states_list[l][:, :n], self.streaming_value_states_list[l][:, :n]

    def compress_and_replace_streaming_kv(self, l, sk, sv):
        n, W = sk.shape[1], self.sink_size + self.recent_size
        if n <= W:
            self.streaming_key_states_list[l][:, :n].copy_(sk)
            self.streaming_value_states_list[l][:, :n].copy_(sv)
            self.streaming_kv_seq_len_list[l] = n
        else:
            s, r = self.sink_size, self.recent_size
            self.streaming_key_states_list[l][:, :s].copy_(sk[:, :s])
            self.streaming_key_states_list[l][:, s:s + r].copy_(sk[:, n - r:n])
            self.streaming_value_states_list[l][:, :s].copy_(sv[:, :s])
            self.streaming_value_states_list[l][:, s:s + r].copy_(sv[:, n - r:n])
            self.streaming_kv_seq_len_list[l] = W

    def clear(self):
        for l in range(self.num_layers):
            self.kv_seq_len_list[l] = 0
            self.streaming_kv_seq_len_list[l] = 0

    def evict_last(self, n):
        for l in range(self.num_layers):
            self.kv_seq_len_list[l] = max(0, self.kv_seq_len_list[l] - n)
            self.streaming_kv_seq_len_list[l] = max(0, self.streaming_kv_seq_len_list[l] - n)


def static_forward_ref(q, k, v, cache: StaticCacheRef, layer_idx: int, pos0: int, rope_scale: float,
                       rope_theta: float, round_p=True, out_dtype=None, return_budget=False):
    """Post-projection part of llama_duo_attention_forward_one_way_reordered_static
    (reference llama.py:309-434).  q [B,S,Hq,D], k/v [B,S,Hkv,D] (pre-RoPE; rotated in place like
    the reference).  Returns attn_output [B,S,Hq,D] before o_proj (and the P-rounding error budget
    of flash_attn_func_ref when ``return_budget``)."""
    B, S, Hq, D = q.shape
    Hkv = k.shape[2]
    G = Hq // Hkv
    kv_seq_len = S + cache.kv_seq_len
    apply_rope_inplace_ref(q, k, pos0, rope_scale, rope_theta)
    fk, fv, sk, sv = cache.split_kv(layer_idx, k, v)
    fk, fv = cache.put_full_kv(layer_idx, fk, fv)
    kw = dict(causal=True, round_p=round_p, out_dtype=out_dtype, return_budget=True)
    if S == kv_seq_len:
        out, bud = flash_attn_func_ref(q, k, v, **kw)
    else:
        nfq = cache.num_full_kv_head_list[layer_idx] * G
        ck, cv = cache.get_streaming_kv(layer_idx)
        sk = torch.cat([ck, sk], dim=1)
        sv = torch.cat([cv, sv], dim=1)
        outs = []
        if nfq > 0:
            outs.append(flash_attn_func_ref(q[:, :, :nfq], fk, fv, **kw))
        if Hq - nfq > 0:
            outs.append(flash_attn_func_ref(q[:, :, nfq:], sk, sv, **kw))
        out = outs[0][0] if len(outs) == 1 else torch.cat([o[0] for o in outs], dim=2)
        bud = outs[0][1] if len(outs) == 1 else torch.cat([o[1] for o in outs], dim=2)
    cache.compress_and_replace_streaming_kv(layer_idx, sk, sv)
    return (out, bud) if return_budget else out


def tuple_forward_ref(q, k, v, past: Optional[Tuple[torch.Tensor, torch.Tensor]], nf: int, sink: int,
                      recent: int, round_p=True, out_dtype=None, return_budget=False):
    """Post-RoPE part of llama_duo_attention_forward_one_way_reordered (reference llama.py:146-306).
    q [B,S,Hq,D], k/v [B,S,Hkv,D] already rotated.  past = (full_KV [2B,nf,N,D],
    streaming_KV [2B,ns,n,D]) or None.  Returns (attn_output [B,S,Hq,D], new past) — plus the P-rounding
    error budget of flash_attn_func_ref when ``return_budget``."""
    B, S, Hq, D = q.shape
    Hkv = k.shape[2]
    G = Hq // Hkv
    kv_seq_len = S + (past[0].shape[2] if past is not None else 0)
    fk, fv, sk, sv = k[:, :, :nf], v[:, :, :nf], k[:, :, nf:], v[:, :, nf:]
    if past is not None:
        pf, ps = past[0].transpose(1, 2), past[1].transpose(1, 2)
        fk = torch.cat([pf[:B], fk], dim=1)
        fv = torch.cat([pf[B:], fv], dim=1)
        sk = torch.cat([ps[:B], sk], dim=1)
        sv = torch.cat([ps[B:], sv], dim=1)
    kw = dict(causal=True, round_p=round_p, out_dtype=out_dtype, return_budget=True)
    if S == kv_seq_len:
        out, bud = flash_attn_func_ref(q, k, v, **kw)
    else:
        outs = []
        if nf > 0:
            outs.append(flash_attn_func_ref(q[:, :, :nf * G], fk, fv, **kw))
        if Hkv - nf > 0:
            outs.append(flash_attn_func_ref(q[:, :, nf * G:], sk, sv, **kw))
        out = outs[0][0] if len(outs) == 1 else torch.cat([o[0] for o in outs], dim=2)
        bud = outs[0][1] if len(outs) == 1 else torch.cat([o[1] for o in outs], dim=2)
    if sk.shape[1] > sink + recent:
        sk = torch.cat([sk[:, :sink], sk[:, -recent:]], dim=1)
        sv = torch.cat([sv[:, :sink], sv[:, -recent:]], dim=1)
    new_past = (torch.cat([fk, fv], dim=0).transpose(1, 2), torch.cat([sk, sv], dim=0).transpose(1, 2))
    return (out, new_past, bud) if return_budget else (out, new_past)


def rmsnorm_ref(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    xf = x.float()
    y = xf * torch.rsqrt(xf.pow(2).mean(dim=-1, keepdim=True) + eps) * w.float()
    return y.to(x.dtype)


def rmsnorm_hf_ref(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """transformers LlamaRMSNorm / MistralRMSNorm.forward (what the tuple path's decoder layer runs, reference
    tuple_kv_cache.py:431-490 leaves the norm modules alone): the normalised activations are rounded to the input dtype
    BEFORE the multiplication by the weight, which rounds again."""
    xf = x.float()
    xf = xf * torch.rsqrt(xf.pow(2).mean(dim=-1, keepdim=True) + eps)
    return w * xf.to(x.dtype)


def tuple_decode_prep_ref(q, k, v, cos_row, sin_row, n_full, arena, full_len, str_src, sink, recent):
    """The data movement of llama_duo_attention_forward_one_way_reordered at q_len == 1, one batch row (reference
    llama.py:177-184 rotary, :202-223 concatenations, :273-301 truncation + K-on-V stacks) with the reference's torch ops.
    q [Hq, D], k / v [Hkv, D]: q and k are rotated IN PLACE; ``arena`` [2, nf, cap, D] receives the retrieval heads' rows at
    row ``full_len``; returns the new streaming cache [2, ns, min(n + 1, sink + recent), D] built from ``str_src``
    [2, ns, n, D] — same contract as the product's ``tuple_decode_prep`` backend call."""
    def rot(x):                      # transformers rotate_half + apply_rotary_pos_emb, dtype arithmetic of x
        x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
        return (x * cos_row) + (torch.cat((-x2, x1), dim=-1) * sin_row)

    q.copy_(rot(q))
    k.copy_(rot(k))
    if n_full > 0:
        arena[0, :, full_len].copy_(k[:n_full])
        arena[1, :, full_len].copy_(v[:n_full])
    sk = torch.cat([str_src[0].transpose(0, 1), k[n_full:].unsqueeze(0)], dim=0)      # [n + 1, ns, D]
    sv = torch.cat([str_src[1].transpose(0, 1), v[n_full:].unsqueeze(0)], dim=0)
    if sk.shape[0] > sink + recent:
        sk = torch.cat([sk[:sink], sk[sk.shape[0] - recent:]], dim=0)[: sink + recent]
        sv = torch.cat([sv[:sink], sv[sv.shape[0] - recent:]], dim=0)[: sink + recent]
    return torch.stack([sk, sv], dim=0).transpose(1, 2).contiguous()


def token_linear_ref(x, blocks, norm=None, x2=None, residual=None, exact=False, norm_hf=False):
    """The module sequence either side of the attention op at q_len == 1, written module by module (reference
    llama.py:332-340 q/k/v_proj, :430-432 o_proj; static_kv_cache.py:482-537 norms, MLP, residual adds; HF LlamaMLP
    ``down_proj(act_fn(gate_proj(h)) * up_proj(h))``), every intermediate a bf16 tensor as in the modules:
        xn = x | rmsnorm(x; *norm) | silu(x) * x2;   y = cat_i(xn @ W_i^T + b_i);   y = y + residual
    x [rows, n_in]; blocks: list of (weight [n, n_in], bias or None).  fp64 accumulation of the products (the check's
    tolerance is the summation-order noise of an fp32 dot product).  ``exact``: also return the fp64 value of the product
    before its rounding (for the tolerance)."""
    dt = x.dtype
    if norm is not None:
        xn = (rmsnorm_hf_ref if norm_hf else rmsnorm_ref)(x, norm[0], norm[1])
    elif x2 is not None:
        xn = torch.nn.functional.silu(x.float()).to(dt)          # act_fn output: a bf16 tensor
        xn = (xn.float() * x2.float()).to(dt)                    # times up_proj's output: a bf16 tensor
    else:
        xn = x
    outs = []
    for w, b in blocks:
        o = xn.double() @ w.double().t()
        if b is not None:
            o = o + b.double()
        outs.append(o)
    pre = torch.cat(outs, -1)
    y = pre.float().to(dt)
    if residual is not None:
        y = (residual.float() + y.float()).to(dt)
    return (y, pre) if exact else y


# ----------------------------------------------------------------------------- host-side helpers
def reorder_rows_ref(weight: torch.Tensor, heads: torch.Tensor, repeat: int, channel: str) -> torch.Tensor:
    """reference patch/utils.py:7-34 (boolean-mask permutation, retrieval heads first)."""
    mask = torch.repeat_interleave(heads, repeats=repeat) > 0.5
    if channel == "in":
        return torch.cat([weight[:, mask], weight[:, ~mask]], dim=1)
    return torch.cat([weight[mask, :], weight[~mask, :]], dim=0)


def sparsify_ref(heads: np.ndarray, sparsity: float, rng_uniform: np.ndarray):
    """reference utils.py:353-373 with the tie-break noise passed in explicitly."""
    h = heads + rng_uniform
    thr = np.quantile(h, sparsity)
    if sparsity >= 1:
        thr = 2
    if sparsity <= 0:
        thr = -1
    out = (h >= thr).astype(float)
    return out, 1 - np.mean(out)


# ----------------------------------------------------------------------------- checker backend
class OracleBackend:
    """The product's backend protocol (duo_attn/backend.py) implemented with the oracle, so the CPU
    test-suite can drive the host plumbing without a GPU.  Tests plug it in through
    ``duo_attn.backend._set_backend_for_testing``; the product never does."""

    name = "oracle"

    def __init__(self, round_p=True):
        self.round_p = round_p

    def rope_inplace(self, q, k, pos0, rope_scale, rope_theta):
        q.copy_(rope_ref(q, pos0, rope_scale, rope_theta))
        k.copy_(rope_ref(k, pos0, rope_scale, rope_theta))

    def kv_append(self, k_src, v_src, k_pool, v_pool, dst_row0):
        n = k_src.shape[0]
        k_pool[dst_row0:dst_row0 + n].copy_(k_src)
        v_pool[dst_row0:dst_row0 + n].copy_(v_src)

    def stream_compress(self, k_pool, v_pool, k_new, v_new, cur_len, sink, recent):
        W = sink + recent
        xk = torch.cat([k_pool[:cur_len], k_new], dim=0)
        xv = torch.cat([v_pool[:cur_len], v_new], dim=0)
        T = xk.shape[0]
        if T <= W:
            k_pool[:T].copy_(xk)
            v_pool[:T].copy_(xv)
            return T
        k_pool[:sink].copy_(xk[:sink])
        k_pool[sink:W].copy_(xk[T - recent:])
        v_pool[:sink].copy_(xv[:sink])
        v_pool[sink:W].copy_(xv[T - recent:])
        return W

    def attention(self, q, out, group, full, stream, scale):
        for desc in (full, stream):
            if desc is None or desc[0] <= 0:
                continue
            n_kv, q_off, a, b = desc
            ks = [t[0] for t in (a, b) if t is not None and t[0].shape[0] > 0]
            vs = [t[1] for t in (a, b) if t is not None and t[0].shape[0] > 0]
            kk, vv = torch.cat(ks, dim=0), torch.cat(vs, dim=0)
            qq = q[:, q_off:q_off + n_kv * group]
            o = flash_attn_func_ref(qq[None], kk[None], vv[None], causal=True, softmax_scale=scale,
                                    round_p=self.round_p)
            out[:, q_off:q_off + n_kv * group].copy_(o[0])

    def rmsnorm(self, x, weight, eps):
        return rmsnorm_ref(x, weight, eps)

    def tuple_decode_prep(self, q, k, v, cos_row, sin_row, n_full, arena, full_len, str_src, sink, recent):
        return tuple_decode_prep_ref(q, k, v, cos_row, sin_row, n_full, arena, full_len, str_src, sink, recent)
