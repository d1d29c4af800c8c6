# round 3: full GPU suite + smoke() on the final HEAD (after the fused-layer eligibility hardening)
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r3final2
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
