# round 3, GPU call 32: fused-layer eligibility hardening (attention forward identity, alignment): the model-level tests that take the fused path
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r3af
mkdir -p $O
timeout 900 python -m pytest tests/test_token_linear_gpu.py tests/test_golden_and_model_gpu.py "tests/test_sharded_models_gpu.py::test_tp2_on_hip_equals_single_process_hip" -x -q -k "fused or static_path or graph or tp2" 2>&1 | tail -5 > $O/pytest.txt; cat $O/pytest.txt
timeout 600 python tools/benchmark_static.py --graph --max_length 32768 --prefill_steps 1 --prefill_warmup 0 --decode_steps 50 --decode_warmup 10 --also_module_by_module 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('32K graph decode fused / module by module:', d['avg_generation_time_ms'], d['avg_generation_time_module_by_module_ms'])" | tee $O/decode32k.txt
