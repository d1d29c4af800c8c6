# round 3, GPU call 17: silu*mul pass on prefill chunks + fused decode layers through the model-level test files; whole-model bench
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r3q
mkdir -p $O
timeout 1500 python -m pytest tests/test_token_linear_gpu.py tests/test_golden_and_model_gpu.py tests/test_sharded_models_gpu.py tests/test_tuple_path_gpu.py tests/test_int4_model_gpu.py -x -q 2>&1 | tail -8 > $O/pytest_models.txt; cat $O/pytest_models.txt
timeout 900 python tools/benchmark_static.py --graph --also_module_by_module --prefill_steps 2 --prefill_warmup 1 --decode_steps 100 --decode_warmup 20 2>/dev/null | tail -1 > $O/model.json
python -c "import json; d=json.load(open('$O/model.json')); print({k: d[k] for k in ('prefill_tok_s','avg_context_time_ms','avg_generation_time_ms','avg_generation_time_module_by_module_ms')})"
timeout 900 python tools/benchmark_static.py --graph --sparsity 0 --prefill_steps 1 --prefill_warmup 0 --decode_steps 100 --decode_warmup 20 2>/dev/null | tail -1 > $O/model_full.json
python -c "import json; d=json.load(open('$O/model_full.json')); print('all heads full:', {k: d[k] for k in ('prefill_tok_s','avg_generation_time_ms')})"
