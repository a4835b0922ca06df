echo "== default"; python -m pytest tests -m gpu -x -q 2>&1 | tail -1
echo "== M3L_ENC_MEGA=3 M3L_DROP_H=1"; M3L_ENC_MEGA=3 M3L_DROP_H=1 python -m pytest tests/test_fulldepth_gpu.py tests/test_parity_gpu.py -x -q 2>&1 | tail -1
echo "== M3L_WGRAD_INLINE=1 M3L_T192=0"; M3L_WGRAD_INLINE=1 M3L_T192=0 python -m pytest tests/test_fulldepth_gpu.py tests/test_parity_gpu.py -x -q 2>&1 | tail -1
echo "== M3L_ATTN_BLOCK=0"; M3L_ATTN_BLOCK=0 python -m pytest tests/test_fulldepth_gpu.py -x -q 2>&1 | tail -1
echo "== rehearse 2 ranks"; M3L_BENCH_REHEARSE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2>&1 | tail -2 | cut -c1-400
