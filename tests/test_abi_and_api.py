"""The C-ABI library loads without a GPU and exports every symbol the header declares; the Python
package exposes the reference's public surface.  No compute calls here."""
import ctypes
import inspect
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "duo_attn_hip.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(duo_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_header_symbols_are_exported_and_typed():
    from duo_attn import _hip

    lib = _hip.load_library()
    declared = _declared_functions()
    assert len(declared) >= 12
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/duo_attn_hip.h but not exported"
    assert set(declared) == set(_hip.EXPORTED_SYMBOLS), "ctypes binding and header disagree"
    assert lib.duo_abi_version() == _hip.ABI_VERSION
    assert lib.duo_target_arch() == b"gfx950"
    assert b"head_dim" in lib.duo_error_string(-2)


def test_struct_layout_matches_header(tmp_path):
    """every struct that crosses the boundary: sizeof and every field offset of the ctypes mirror == what a C compiler
    makes of include/duo_attn_hip.h (a probe program compiled with gcc against the header itself)"""
    import shutil
    import subprocess

    from duo_attn import _hip

    pairs = {"duo_kv_seg": _hip.KVSeg, "duo_head_class": _hip.HeadClass, "duo_int4_pool": _hip.Int4Pool,
             "duo_decode_layer_args": _hip.DecodeLayerArgs, "duo_decode_batch": _hip.DecodeBatch,
             "duo_linear_seg": _hip.LinearSeg, "duo_token_linear_args": _hip.TokenLinearArgs,
             "duo_tuple_decode_args": _hip.TupleDecodeArgs}
    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void) {"]
    for cname, cls in pairs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines.append("return 0; }")
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.run([gcc, "-std=c11", "-o", str(exe), str(src)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in pairs.items():
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"
    assert ctypes.sizeof(_hip.KVSeg) == 48 and _hip.HeadClass.segB.offset == 8 + 48      # ABI v2: + batch_stride


def test_argument_errors_without_gpu():
    """argument validation happens on the host before any launch"""
    from duo_attn import _hip

    lib = _hip.load_library()
    assert lib.duo_attn_decode_workspace_bytes(32, 512) == 32 * 512 * 130 * 4
    # ABI v5: the decode entry points size their split-KV grid from the bucket of the visible rows — 64-token units rounded
    # up to a power of two (host arithmetic; duo_attn/graph.py re-captures a decode step when it changes)
    assert [lib.duo_decode_plan_bucket(n) for n in (-3, 0, 1, 64, 65, 128, 129, 256, 257, 131072, 131073, 3300000)] == \
        [0, 0, 1, 1, 2, 2, 4, 4, 8, 2048, 4096, 65536]
    for n in range(1, 5000, 37):
        b = lib.duo_decode_plan_bucket(n)
        units = (n + 63) // 64
        assert b & (b - 1) == 0 and units <= b < 2 * units
    rc = lib.duo_rope_inplace_bf16(None, 0, 0, 0, None, 0, 0, 0, 4, 0, 1.0, 1e4, 64, None)
    assert rc == -2   # DUO_EHEADDIM
    rc = lib.duo_attn_prefill_bf16(None, 0, 0, None, 0, 0, 4, 4, None, None, 1.0, 128, None)
    assert rc == -1   # DUO_EINVAL (null q)
    a = _hip.TokenLinearArgs()
    assert lib.duo_token_linear_bf16(ctypes.byref(a), None) == -1      # null x / y
    assert _hip.TOKEN_LINEAR_MAX_ROWS == 4
    # one padding rule for the LDS limit, shared by the header, the kernel and the Python gate
    hdr = open(HEADER).read()
    assert int(re.search(r"#define DUO_TOKEN_LINEAR_PAD (\d+)", hdr).group(1)) == _hip.TOKEN_LINEAR_PAD == 2048
    assert int(re.search(r"#define DUO_LINEAR_NORM_HF (\d+)", hdr).group(1)) == _hip.LINEAR_NORM_HF
    assert _hip.token_linear_fits(4, 14336) and not _hip.token_linear_fits(4, 20480 + 8) and _hip.token_linear_fits(1, 4096)
    t = _hip.TupleDecodeArgs()
    assert lib.duo_tuple_decode_prep_bf16(ctypes.byref(t), None, None) == -2       # head_dim 0: DUO_EHEADDIM
    assert lib.duo_attn_decode_int4_f16(None, 0, None, 0, 4, None, None, 1.0, 128, 7, None, 0, None) == -1     # bad q / fused


def test_missing_library_fails_loudly(tmp_path):
    from duo_attn import _hip

    with pytest.raises(_hip.DuoHipError, match="no CPU fallback"):
        _hip.load_library(str(tmp_path / "nope.so"))


def test_cpu_tensors_are_refused():
    from duo_attn import _hip

    q = torch.zeros(2, 4, 128, dtype=torch.bfloat16)
    with pytest.raises(_hip.DuoHipError, match="no CPU fallback"):
        _hip.rope_inplace(q, q, 0, 1.0, 1e4)
    with pytest.raises(_hip.DuoHipError, match="no CPU fallback"):
        _hip.attn_prefill(q, torch.empty_like(q), 1, None, None, 1.0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "duo-attention_amd", "duo_attn")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f


def test_public_api_surface():
    import duo_attn.patch as P
    import duo_attn.patch.llama as L
    import duo_attn.patch.mistral as M
    import duo_attn.patch.tuple_kv_cache as T
    import duo_attn.utils as U

    for n in ("enable_duo_attention_eval", "enable_duo_attention_training", "get_full_attention_heads",
              "set_full_attention_heads", "map_full_attention_heads", "load_full_attention_heads"):
        assert callable(getattr(P, n))
    assert list(inspect.signature(P.enable_duo_attention_eval).parameters) == [
        "model", "full_attention_heads", "sink_size", "recent_size"]
    for mod, fam in ((L, "llama"), (M, "mistral")):
        for n in (f"enable_{fam}_duo_attention_static_kv_cache_eval", f"enable_{fam}_duo_attention_eval",
                  "DuoAttentionStaticKVCache", f"get_{fam}_full_attention_heads"):
            assert hasattr(mod, n), n
    assert list(inspect.signature(L.DuoAttentionStaticKVCache.__init__).parameters)[1:] == [
        "model", "full_attention_heads", "batch_size", "max_size", "sink_size", "recent_size"]
    for n in ("kv_seq_len", "streaming_kv_seq_len", "split_kv", "put_full_kv", "get_streaming_kv",
              "compress_and_replace_streaming_kv", "clear", "evict_last", "memory_usage", "get_full_kv"):
        assert hasattr(L.DuoAttentionStaticKVCache, n), n
    assert callable(T.enable_tuple_kv_cache)
    for n in ("load_attn_pattern", "sparsify_attention_heads", "seed_everything", "get_model", "get_tokenizer",
              "to_device", "parse_args", "save_full_attention_heads"):
        assert callable(getattr(U, n)), n
    with pytest.raises(NotImplementedError):
        P.enable_duo_attention_training(type("M", (), {"config": type("C", (), {"model_type": "llama"})})(), 1, 1, 1)
    with pytest.raises(ValueError, match="not supported"):
        P.enable_duo_attention_eval(type("M", (), {"config": type("C", (), {"model_type": "gpt2"})})(), [], 1, 1)


def test_parse_args_keeps_reference_flags():
    from duo_attn.utils import parse_args

    a = parse_args(["--model_name", "m", "--attn_load_dir", "d", "--sparsity", "0.5", "--max_length", "100000",
                    "--prefilling_chunk_size", "32000", "--device", "0", "--seed", "42"])
    assert a.sparsity == 0.5 and a.prefilling_chunk_size == 32000 and a.device == "cuda:0"
    assert a.sink_size == 64 and a.recent_size == 256


def test_decode_layer_struct_layout():
    """ctypes mirror of duo_decode_layer_args == the C struct (offsets checked against gcc's layout of
    include/duo_attn_hip.h)."""
    import subprocess
    import tempfile

    from duo_attn import _hip

    src = r'''
#include "%s"
#include <stdio.h>
#include <stddef.h>
int main() {
  printf("%%zu %%zu %%zu %%zu %%zu %%zu\n", sizeof(duo_decode_layer_args), offsetof(duo_decode_layer_args, k),
         offsetof(duo_decode_layer_args, full_k), offsetof(duo_decode_layer_args, str_len),
         offsetof(duo_decode_layer_args, pos), offsetof(duo_decode_layer_args, scale));
  return 0;
}''' % HEADER
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        subprocess.run(["gcc", c, "-o", os.path.join(d, "t")], check=True)
        got = [int(x) for x in subprocess.run([os.path.join(d, "t")], capture_output=True, text=True, check=True).stdout.split()]
    A = _hip.DecodeLayerArgs
    assert got == [ctypes.sizeof(A), A.k.offset, A.full_k.offset, A.str_len.offset, A.pos.offset, A.scale.offset]


def test_pattern_files_round_trip(tmp_path):
    """save_full_attention_heads -> load_full_attention_heads / load_attn_pattern (reference utils.py:326-350,
    patch/__init__.py:110-121): the TSV + config.json layout the shipped attn_patterns use."""
    import json

    import numpy as np
    from duo_attn.patch import load_full_attention_heads
    from duo_attn.utils import load_attn_pattern, save_full_attention_heads, sparsify_attention_heads

    heads = np.random.RandomState(0).rand(4, 8)
    save_full_attention_heads(torch.tensor(heads), str(tmp_path / "full_attention_heads.tsv"))
    (tmp_path / "config.json").write_text(json.dumps({"sink_size": 64, "recent_size": 256}))
    got, sink, recent = load_attn_pattern(str(tmp_path))
    assert (sink, recent) == (64, 256) and got.shape == (4, 8)
    assert np.allclose(got, heads, atol=1e-6)
    again = np.asarray(load_full_attention_heads(str(tmp_path)))
    assert np.allclose(again, heads, atol=1e-6)
    binar, sp = sparsify_attention_heads(got.copy(), None, 0.5)
    assert set(np.unique(binar)) <= {0.0, 1.0} and abs(sp - 0.5) < 0.07


def test_to_device_contract():
    """single device passes through; a device list without a mode is an error; TP / PP are one process per GPU"""
    from duo_attn.utils import to_device

    m = torch.nn.Linear(2, 2)
    assert to_device(m, "cpu") is m
    with pytest.raises(ValueError):
        to_device(m, [0, 1])
    with pytest.raises(RuntimeError, match="one process per GPU"):
        to_device(m, [0, 1], enable_tp=True)          # (the sharding itself: tests/test_tp_gloo.py)


def test_generated_prefill_schedule_is_in_sync(tmp_path):
    """duo_prefill_w64_bulk.inc is generated (tools/gen_w64_bulk.py): the committed file must be what the committed
    generator writes with its default switches."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "bulk.inc"
    env = {k: v for k, v in os.environ.items() if not k.startswith("W64_GEN_")}
    subprocess.run([sys.executable, os.path.join(root, "tools", "gen_w64_bulk.py"), "-o", str(out)], check=True, env=env)
    committed = open(os.path.join(root, "duo-attention_amd", "csrc", "duo_prefill_w64_bulk.inc")).read()
    assert out.read_text() == committed
