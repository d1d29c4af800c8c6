"""CPU: the host-side decisions of the automatic decode graph (duo_attn/graph.py) that need no GPU — what retires a captured
step (the storage of any parameter or buffer, a replaced module) and what keeps a model eager (a forward hook anywhere)."""
import torch


def _tiny():
    from transformers import LlamaConfig, LlamaForCausalLM

    torch.manual_seed(0)
    return LlamaForCausalLM(LlamaConfig(hidden_size=128, intermediate_size=64, num_hidden_layers=2, num_attention_heads=1,
                                        num_key_value_heads=1, head_dim=128, vocab_size=32, tie_word_embeddings=False))


def test_walk_sees_every_parameter_buffer_and_hook():
    from duo_attn import graph

    m = _tiny()
    hooked, ptrs = graph._walk(m)
    n_params, n_bufs = sum(1 for _ in m.parameters()), sum(1 for _ in m.buffers())
    assert not hooked and len(ptrs) == n_params + n_bufs and n_bufs >= 1
    assert {p.data_ptr() for p in m.parameters()} <= set(ptrs) and {b.data_ptr() for b in m.buffers()} <= set(ptrs)
    # a partial .data swap of ONE weight that is neither q_proj nor down_proj
    w = m.model.layers[1].self_attn.k_proj.weight
    w.data = w.data.clone()
    assert graph._walk(m)[1] != ptrs
    ptrs = graph._walk(m)[1]
    # a replaced module (new object, new parameter): the tree is walked per call, nothing is cached
    m.lm_head = torch.nn.Linear(128, 32, bias=False)
    assert graph._walk(m)[1] != ptrs
    # a re-pointed buffer
    ptrs = graph._walk(m)[1]
    name, buf = next(iter(m.named_buffers()))
    owner = m.get_submodule(name.rsplit(".", 1)[0]) if "." in name else m
    owner._buffers[name.rsplit(".", 1)[-1]] = buf.clone()
    assert graph._walk(m)[1] != ptrs
    # hooks: on a leaf, on a container, as a pre-hook, globally
    for mod, reg in ((m.model.layers[0].mlp.gate_proj, "register_forward_hook"), (m.model, "register_forward_pre_hook")):
        h = getattr(mod, reg)(lambda *a: None)
        assert graph._walk(m)[0]
        h.remove()
        assert not graph._walk(m)[0]
    h = torch.nn.modules.module.register_module_forward_hook(lambda *a: None)
    try:
        assert graph._walk(m)[0]
    finally:
        h.remove()
    assert not graph._walk(m)[0]


def test_modules_hooked_guards_the_bypassing_forms():
    from duo_attn.patch._duo import modules_hooked

    m = _tiny()
    mlp = m.model.layers[0].mlp
    assert not modules_hooked((mlp, mlp.act_fn, m.model.layers[0].input_layernorm))
    h = mlp.register_forward_hook(lambda *a: None)
    assert modules_hooked((m.model.layers[0].input_layernorm, mlp)) and not modules_hooked((m.model.layers[1].mlp,))
    h.remove()
    assert not modules_hooked((mlp,))
    assert not modules_hooked((lambda x: x, object()))         # (an activation that is a plain function, a foreign object)


def test_signature_changes_with_forwards_and_weights():
    from duo_attn import graph

    m = _tiny()
    s0 = graph._model_signature(m)
    assert graph._model_signature(m) == s0
    m.model.layers[0].post_attention_layernorm.weight.data = m.model.layers[0].post_attention_layernorm.weight.data.clone()
    s1 = graph._model_signature(m)
    assert s1 != s0
    import types

    m.model.layers[1].forward = types.MethodType(lambda self, *a, **k: None, m.model.layers[1])
    assert graph._model_signature(m) != s1
