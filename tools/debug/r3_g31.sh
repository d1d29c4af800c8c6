# round 3, GPU call 31: rehearsal of the N > 1 paths on the final tree (2 ranks on cuda:0, gloo): bench.py --gpus 2, benchmark_static --pp / --tp
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r3ae
mkdir -p $O
export DUO_BENCH_DEBUG_SHARED_GPU=1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 1 --warmup 0 --ctx 32768 --chunk 8192 --layers 8 --decode-tokens 8 --no-full-baseline 2>$O/n2.err | tail -1 > $O/n2_rehearsal.json
python -c "import json; d=json.load(open('$O/n2_rehearsal.json')); print({k: d[k] for k in ('n_gpus','value','scaling')}, d['pipeline']['backend'], d['roofline_token_linear'], d['model_level'])"; tail -2 $O/n2.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 tools/benchmark_static.py --pp --row_block 2048 --max_length 16384 --prefilling_chunk_size 8192 --prefill_steps 1 --prefill_warmup 0 --decode_steps 8 --decode_warmup 2 2>$O/pp.err | tail -1 > $O/pp.json; cat $O/pp.json | cut -c1-400; tail -2 $O/pp.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29532 tools/benchmark_static.py --tp --max_length 16384 --prefilling_chunk_size 8192 --prefill_steps 1 --prefill_warmup 0 --decode_steps 8 --decode_warmup 2 2>$O/tp.err | tail -1 > $O/tp.json; cat $O/tp.json | cut -c1-400; tail -2 $O/tp.err
