# round 3, GPU call 33: the reference's decoder-layer golden on the HIP path
cd $GRAFT_REPO_ROOT
timeout 200 python -m pytest tests/test_token_linear_gpu.py -q -s -k "reference_golden" 2>&1 | grep "layer_a\|passed\|failed\|Error" | tail -20
