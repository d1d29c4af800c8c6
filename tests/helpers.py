"""Shared test helpers (shape-only model stub, tolerance checks)."""
import types

import torch


class ShapeModel:
    """What DuoAttentionStaticKVCache reads from a model: config dims + device/dtype of parameters."""

    def __init__(self, num_layers, num_heads, num_kv_heads, head_dim=128, device="cpu", dtype=torch.bfloat16):
        self.config = types.SimpleNamespace(
            num_hidden_layers=num_layers,
            num_attention_heads=num_heads,
            num_key_value_heads=num_kv_heads,
            hidden_size=num_heads * head_dim,
        )
        self._p = torch.zeros(1, device=device, dtype=dtype)

    def parameters(self):
        yield self._p


def heads_from_counts(counts, num_kv_heads):
    """per-layer [1]*nf + [0]*ns rows (already reordered form)."""
    return [[1.0] * nf + [0.0] * (num_kv_heads - nf) for nf in counts]


# measured error of every attn_close call of the session (what -> numbers); tests/conftest.py writes it to
# gpurun_out/parity_report.json at the end of a GPU run, and the round's copy is committed as profiles/parity_rNN.json
PARITY_LOG = {}


NOISE_RATIO_BAR = 1.12      # see attn_close


def with_rounded(budget: torch.Tensor, ref_rounded: torch.Tensor) -> torch.Tensor:
    """attach the REFERENCE ARITHMETIC's own result on the same inputs (P rounded to the element type before P.V, output
    rounded to the element type — what FA2 computes; the oracle's ``round_p=True`` form) to the budget tensor an
    ``attn_close`` call receives: the call then also holds the kernel to that arithmetic's own noise"""
    budget.ref_rounded = ref_rounded
    return budget


def attn_close(ours: torch.Tensor, ref_exact: torch.Tensor, what="", budget: torch.Tensor = None, ref_rounded: torch.Tensor = None):
    """Parity bar for bf16 attention outputs against the EXACT-P fp32 oracle.

    north_star asks for 1e-3 relative.  Two roundings that the reference's own kernel (FA2) also
    performs are irreducible and are budgeted explicitly rather than hidden in a loose rtol:
      * the bf16 OUTPUT rounding: half an ulp = 2^-9 relative (already > 1e-3); one ulp is allowed;
      * P rounded to bf16 before P.V (prefill/MFMA path only): each p_j carries <= 2^-9 relative
        error, so |dO| <= 2^-9 * sum_j p_j |v_j| =: 2^-9 * budget; 2^-8 * budget is allowed
        (``budget`` comes from the oracle, return_budget=True; None for the fp32-P decode path).
    Elementwise:  |ours - ref| <= 1e-3*|ref| + 2^-8*|ref| + 2^-8*budget + 1e-3*rms(ref)
    Statistical:  rms(ours - ref) <= 2.5e-3 * rms(ref)   (random rounding noise sits near 1.2e-3;
                  a wrong mask bit or a mis-scaled tile is orders of magnitude above it).
    Against the reference arithmetic itself (round 6; ``ref_rounded`` or ``helpers.with_rounded(budget, ...)``): the absolute
    bar above was calibrated on N(0, 1) data, where the noise of the reference's OWN arithmetic — P rounded to bf16 before
    P.V, bf16 output — is 2.33e-3 of rms; on other data that noise moves (2.55e-3 on low-variance scores, profiles/
    r5_fuzz.md), and a kernel well noisier than the reference would still pass.  So where the rounded form is available:
                  rms(ours - ref) <= (NOISE_RATIO_BAR + 3 / sqrt(n)) * rms(ref_rounded - ref)
    (two independent draws of the same noise over n elements differ by ~1 / sqrt(n) in rms); when that holds and every
    element passed, the absolute bar may be crossed by the same margin the reference arithmetic crosses it.
    NOISE_RATIO_BAR is not 1.00: the rounded ORACLE rounds P relative to the row's TRUE maximum, so the row's largest weight
    is exactly 1.0 and carries no rounding error at all; a streaming kernel (FA2 too) rounds P relative to the running maximum
    of the tiles seen so far — here raised only when a score exceeds it by 2^8 — so that weight is a number like 1.37 and is
    rounded like every other one.  Measured on the MI355X over the suite (profiles/parity_r6.json,
    ``rms_err_over_reference_arithmetic``): 1.02 - 1.04 on ordinary launches, up to 1.09 on launches whose rows see a handful
    of keys (one dominant weight per row), 0.97 - 1.03 when the key range is split (another draw, not a better kernel).
    """
    o = ours.float().cpu()
    r = ref_exact.float().cpu()
    assert o.shape == r.shape, (o.shape, r.shape)
    assert torch.isfinite(o).all(), f"{what}: non-finite output"
    err = (o - r).abs()
    rms = r.pow(2).mean().sqrt()
    tol = 1e-3 * r.abs() + (2.0 ** -8) * r.abs() + 1e-3 * rms
    if budget is not None:
        tol = tol + (2.0 ** -8) * budget.float().cpu()
    err_rms = (o - r).pow(2).mean().sqrt()
    if ref_rounded is None and budget is not None:
        ref_rounded = getattr(budget, "ref_rounded", None)
    ref_noise = None
    if ref_rounded is not None:
        rr = ref_rounded.float().cpu()
        assert rr.shape == r.shape, (rr.shape, r.shape)
        ref_noise = (rr - r).pow(2).mean().sqrt()
    if what and o.numel():
        PARITY_LOG[what] = {
            "n": int(o.numel()),
            "max_abs_err": float(err.max()),
            "rms_err": float(err_rms),
            "rms_ref": float(rms),
            "rms_err_over_rms_ref": float(err_rms / rms) if float(rms) > 0 else 0.0,
            "rms_bar": 2.5e-3,
            "worst_err_over_elementwise_tol": float((err / tol.clamp_min(1e-30)).max()),
            "p_rounding_budget": budget is not None,
        }
        if ref_noise is not None:
            PARITY_LOG[what]["rms_err_of_reference_arithmetic"] = float(ref_noise)
            PARITY_LOG[what]["rms_err_over_reference_arithmetic"] = float(err_rms / ref_noise) if float(ref_noise) > 0 else 0.0
    bad = err > tol
    assert not bad.any(), (
        f"{what}: {int(bad.sum())}/{bad.numel()} elements out of tolerance; max err {err.max():.3e} "
        f"at ref {r.flatten()[err.argmax()]:.3e}, rms {rms:.3e}"
    )
    if ref_noise is not None and float(ref_noise) > 0:
        lim = (NOISE_RATIO_BAR + 3.0 / max(1, o.numel()) ** 0.5) * ref_noise
        assert err_rms <= lim, (f"{what}: rms err {err_rms:.3e} is more than {NOISE_RATIO_BAR - 1:.0%} above the reference arithmetic's own "
                                f"{ref_noise:.3e} on the same inputs (rms(ref) {rms:.3e})")
        # the reference arithmetic may itself sit past the absolute bar on this data: the kernel is held to IT
        assert err_rms <= max(2.5e-3 * rms, lim), f"{what}: rms err {err_rms:.3e} vs rms(ref) {rms:.3e}"
        return
    assert err_rms <= 2.5e-3 * rms, f"{what}: rms err {err_rms:.3e} vs rms(ref) {rms:.3e}"


def rel_close(got, want, bar, what):
    """model-level comparison (logits, cache rows) in relative L2; the MEASURED figure goes to the parity report
    (``model: ...`` entries) and, one line per call, to gpurun_out/model_rel.log (worker processes included), so the bars
    can be kept at about twice what is measured instead of at a round number"""
    import os

    g, w = got.float().cpu(), want.float().cpu()
    assert g.shape == w.shape, (what, g.shape, w.shape)
    rel = float((g - w).norm() / w.norm().clamp_min(1e-30))
    PARITY_LOG["model: " + what] = {"rel_l2": rel, "bar": bar, "n": int(g.numel())}
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        try:
            with open(os.path.join(d, "model_rel.log"), "a") as f:
                f.write(f"{rel:.4e}\t{bar:.1e}\t{what}\n")
        except OSError:
            pass
    assert rel < bar, f"{what}: relative L2 {rel:.3e} >= {bar:.1e}"
    return rel


# ----------------------------------------------------------------------------------------------------------------------
# point-to-point discipline of a multi-process run (what gloo cannot show, RCCL would hang on)
# ----------------------------------------------------------------------------------------------------------------------
class P2PAudit:
    """Records every torch.distributed point-to-point call of this process together with the COMMUNICATOR torch's RCCL/NCCL
    process group would run it on when the group was initialised LAZILY (the stricter case): a plain ``send / recv / isend /
    irecv`` uses the two-rank communicator of its pair, the same call inside ``batch_isend_irecv`` the group-wide one
    (ProcessGroupNCCL::pointToPoint: ``batchP2P`` selects the device key, a single operation the send-recv key; with eager
    initialisation — ``init_process_group(device_id=...)`` — plain calls run on the group's communicator too) — and operations
    on different communicators never match.  gloo
    matches by source and tag alone, so a run over gloo cannot reveal a send issued one way and its receive the other;
    ``check_p2p_logs`` over the gathered logs can.

        with P2PAudit() as log: ...run...; logs = [None] * world; dist.all_gather_object(logs, log.calls); check_p2p_logs(logs)
    """

    def __init__(self):
        self.calls = []         # (direction "send" | "recv", global peer rank, "pair" | "group", batch id)
        self._batched = False
        self._inside = False
        self._batch_id = 0

    def __enter__(self):
        import torch.distributed as dist
        import torch.distributed.distributed_c10d as c10d

        self._mods, self._saved = (dist, c10d), {}

        def put(name, fn):
            for m in self._mods:        # both names: P2POp validates its op against distributed_c10d's own globals
                setattr(m, name, fn)

        def wrap(name, direction, peer_kw):
            orig = getattr(c10d, name)
            self._saved[name] = orig

            def f(tensor, *a, **kw):
                peer = kw.get(peer_kw)
                if peer is None:
                    peer = kw.get("group_" + peer_kw, a[0] if a else None)
                    if kw.get("group") is not None and kw.get("group_" + peer_kw) is not None:
                        peer = dist.get_global_rank(kw["group"], peer)
                if self._inside:        # send() / recv() are built on isend() / irecv() in some torch versions: one record
                    return orig(tensor, *a, **kw)
                if not self._batched:
                    self._batch_id += 1         # a plain call is a batch of its own
                self.calls.append((direction, int(peer), "group" if self._batched else "pair", self._batch_id))
                self._inside = True
                try:
                    return orig(tensor, *a, **kw)
                finally:
                    self._inside = False

            f.__name__ = name
            put(name, f)

        for n, d, k in (("send", "send", "dst"), ("isend", "send", "dst"), ("recv", "recv", "src"), ("irecv", "recv", "src")):
            wrap(n, d, k)
        orig_batch = c10d.batch_isend_irecv
        self._saved["batch_isend_irecv"] = orig_batch

        def batch(ops):      # the calls it makes (through the wrapped isend / irecv) are group-communicator operations
            self._batch_id += 1
            self._batched = True
            try:
                return orig_batch(ops)
            finally:
                self._batched = False

        put("batch_isend_irecv", batch)
        return self

    def __exit__(self, *exc):
        for n, o in self._saved.items():
            for m in self._mods:
                setattr(m, n, o)
        return False


def check_p2p_logs(logs):
    """``logs[r]`` = P2PAudit.calls of global rank r.  For every ordered pair (src -> dst): as many sends as receives, and
    the i-th send runs on the same kind of communicator as the i-th receive; pair communicators mirror each other; and the
    batches issued on the GROUP communicator — which runs each rank's batches in issue order, the operations of one batch
    together — can all complete (a simulation of exactly that: stuck = a hang on RCCL)."""
    full = [[(c + (1_000_000 + i,))[:4] for i, c in enumerate(lg)] for lg in logs]      # (3-tuples: every call its own batch)
    logs = [[c[:3] for c in lg] for lg in full]
    world = len(logs)
    n_hops = 0
    for src in range(world):
        for dst in range(world):
            sends = [k for d, p, k in logs[src] if d == "send" and p == dst]
            recvs = [k for d, p, k in logs[dst] if d == "recv" and p == src]
            assert len(sends) == len(recvs), f"{src} -> {dst}: {len(sends)} sends, {len(recvs)} receives"
            for i, (a, b) in enumerate(zip(sends, recvs)):
                assert a == b, (f"hop {src} -> {dst}, operation {i}: the send runs on the {a} communicator, its receive on the {b} "
                                f"communicator — they would never match on RCCL/NCCL")
            n_hops += len(sends)
    # and inside one pair communicator operations run in issue order on BOTH ranks, a send completing only against the
    # receive posted opposite it: the two ranks' sequences on the pair must mirror each other (send facing receive at every
    # position) — two sends facing each other wait for receives queued behind them.  (The autoregressive stream on two
    # stages — items one way, tokens back on the same pair — is the case that exercises it.)
    for a in range(world):
        for b in range(a + 1, world):
            sa = [d for d, p, k in logs[a] if p == b and k == "pair"]
            sb = [d for d, p, k in logs[b] if p == a and k == "pair"]
            assert len(sa) == len(sb), f"pair ({a}, {b}): {len(sa)} vs {len(sb)} operations"
            for i, (x, y) in enumerate(zip(sa, sb)):
                assert x != y, f"pair ({a}, {b}), operation {i}: both ranks {x} — neither can complete on an in-order communicator"
    _simulate_group_communicator(full)                       # lazily initialised group: batched calls on its communicator
    _simulate_group_communicator(full, plain_too=True)       # eagerly initialised group: plain calls run there as well
    return n_hops


def _simulate_group_communicator(logs, plain_too=False):
    """in-order execution of every rank's batches on one communicator: the head batch of a rank retires when each of its
    operations has met its complement (send <-> recv, facing ranks) in the head batch of the peer.  ``plain_too``: plain
    calls are batches of one on the same communicator (what torch does for a group initialised with ``device_id``)"""
    queues = []
    for lg in logs:
        q, last = [], None
        for d, p, k, b in lg:
            if k != "group" and not plain_too:
                continue
            if b != last:
                q.append([])
                last = b
            q[-1].append((d, p))
        queues.append(q)
    progress = True
    while progress:
        progress = False
        for r, q in enumerate(queues):
            while q and not q[0]:
                q.pop(0)
                progress = True
            if not q:
                continue
            for op in list(q[0]):
                d, p = op
                want = ("recv" if d == "send" else "send", r)
                if queues[p] and want in queues[p][0]:
                    q[0].remove(op)
                    queues[p][0].remove(want)
                    progress = True
    stuck = {r: q[0] for r, q in enumerate(queues) if q}
    assert not stuck, f"the group communicator cannot make progress (head batch per rank): {stuck}"
