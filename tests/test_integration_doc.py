"""INTEGRATION.md's reference-side binding stub cannot rot (it asserted ABI v2 against a v3 library).

The ``Structure`` classes and the version assert of the stub are EXTRACTED from the Markdown, executed, and compared with
``include/duo_attn_hip.h`` — through the ctypes mirrors of ``duo_attn/_hip.py``, whose layouts ``tests/test_abi_and_api.py``
checks against a gcc build of the header — and every ``_lib.duo_*`` call the stub makes must be an exported symbol whose
argument count matches the binding's."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOC = os.path.join(ROOT, "INTEGRATION.md")
HEADER = os.path.join(ROOT, "include", "duo_attn_hip.h")


def _python_blocks():
    return re.findall(r"```python\n(.*?)```", open(DOC).read(), flags=re.S)


def _struct_classes():
    """every `class X(Structure): _fields_ = [...]` of the doc's python blocks, executed in one namespace"""
    ns = {}
    exec("import ctypes\nfrom ctypes import c_void_p, c_int32, c_int64, c_float, POINTER, Structure, byref, c_int", ns)
    found = []
    for block in _python_blocks():
        for m in re.finditer(r"^class (\w+)\(Structure\):.*?\n(?:[ \t]+.*\n)+", block, flags=re.M):
            exec(m.group(0), ns)
            found.append(m.group(1))
    return ns, found


def test_stub_version_assert_is_the_headers():
    doc = open(DOC).read()
    asserted = [int(x) for x in re.findall(r"assert _lib\.duo_abi_version\(\) == (\d+)", doc)]
    declared = int(re.search(r"#define DUO_ABI_VERSION (\d+)", open(HEADER).read()).group(1))
    assert asserted and all(a == declared for a in asserted), (asserted, declared)
    from duo_attn import _hip

    assert _hip.ABI_VERSION == declared and _hip.load_library().duo_abi_version() == declared


def test_stub_structures_match_the_binding():
    from duo_attn import _hip

    ns, found = _struct_classes()
    mirror = {"KVSeg": _hip.KVSeg, "HeadClass": _hip.HeadClass, "DecodeLayerArgs": _hip.DecodeLayerArgs,
              "LinearSeg": _hip.LinearSeg, "TokenLinearArgs": _hip.TokenLinearArgs, "Int4Pool": _hip.Int4Pool,
              "TupleDecodeArgs": _hip.TupleDecodeArgs}
    assert set(found) == set(mirror), (sorted(found), sorted(mirror))      # every struct of the doc is checked, none is missing
    for name, ref in mirror.items():
        doc_cls = ns[name]
        assert ctypes.sizeof(doc_cls) == ctypes.sizeof(ref), name
        doc_fields = [(f[0], getattr(doc_cls, f[0]).offset, getattr(doc_cls, f[0]).size) for f in doc_cls._fields_]
        ref_fields = [(f[0], getattr(ref, f[0]).offset, getattr(ref, f[0]).size) for f in ref._fields_]
        assert doc_fields == ref_fields, (name, doc_fields, ref_fields)


def test_stub_calls_exist_with_the_bindings_arity():
    """every `_lib.duo_*(...)` call written out in the doc names an exported symbol and passes as many arguments as the
    ctypes signature takes (calls abbreviated with `…` / `...` are skipped)"""
    from duo_attn import _hip

    doc = "\n".join(_python_blocks())
    calls = 0
    for m in re.finditer(r"_lib\.(duo_\w+)\(", doc):
        name = m.group(1)
        assert name in _hip._SIGNATURES, f"{name} is not an exported symbol"
        depth, i = 1, m.end()
        while depth and i < len(doc):
            depth += doc[i] == "("
            depth -= doc[i] == ")"
            i += 1
        args = doc[m.end():i - 1]
        if "…" in args or "..." in args or not args.strip():
            continue
        n, depth, cur = 0, 0, ""
        for ch in args:
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
            if ch == "," and depth == 0:
                n += 1
                cur = ""
            else:
                cur += ch
        n += 1 if cur.strip() else 0
        assert n == len(_hip._SIGNATURES[name][1]), f"{name}: the doc passes {n} arguments, the ABI takes {len(_hip._SIGNATURES[name][1])}"
        calls += 1
    assert calls >= 5


def test_every_entry_point_named_in_the_doc_is_exported():
    from duo_attn import _hip

    doc = open(DOC).read()
    names = set(re.findall(r"\b(duo_[a-z0-9_]+?)(?:\(|`|\b)", doc))
    entry = {n for n in names if re.match(r"duo_(attn|rope|kv_append|stream_compress|decode_layer|decode_step|decode_state|int4_|token_linear|silu_mul|rmsnorm_bf16|tuple_decode_prep|abi_version|error_string)", n)}
    wildcard = {n for n in entry if n.endswith("_")}        # (`duo_rope_inplace_batched_*` style mentions)
    unknown = sorted(n for n in entry - wildcard
                     if n not in _hip._SIGNATURES and n not in ("duo_int4_pool", "duo_int4_decode_fold_kernel", "duo_attn",
                                                                 "duo_attn_hip", "duo_decode_state", "duo_tuple_decode_args",
                                                                 "duo_decode_layer_args", "duo_token_linear_args",
                                                                 "duo_tuple_decode_layer_fused", "duo_decode_layer_fused"))
    assert not unknown, unknown
