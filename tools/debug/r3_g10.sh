# round 3, GPU call 10: tools/benchmark_static.py --pp / --tp rehearsed on one GPU (gloo, every rank on cuda:0)
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r3j
mkdir -p $O
export DUO_BENCH_DEBUG_SHARED_GPU=1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 tools/benchmark_static.py --pp --row_block 2048 --max_length 16384 --prefilling_chunk_size 8192 --prefill_steps 1 --prefill_warmup 0 --decode_steps 8 --decode_warmup 2 2>$O/pp.err | tail -1 > $O/pp.json; cat $O/pp.json; tail -3 $O/pp.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29532 tools/benchmark_static.py --tp --max_length 16384 --prefilling_chunk_size 8192 --prefill_steps 1 --prefill_warmup 0 --decode_steps 8 --decode_warmup 2 2>$O/tp.err | tail -1 > $O/tp.json; cat $O/tp.json; tail -3 $O/tp.err
