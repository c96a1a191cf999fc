cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus 2 --steps 2 --warmup 1 --rehearse-on-one-gpu > gpurun_out/bench_w2.log 2>&1; echo "exit $?" >> gpurun_out/bench_w2.log)
tail -3 gpurun_out/bench_w2.log | cut -c1-1500
