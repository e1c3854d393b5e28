#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c22; mkdir -p $O
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29641 bench.py --gpus 1 --force-collective --batch 8 --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > $O/torchrun_line.json 2> $O/torchrun.err
echo "rc=$?"
tail -c 900 $O/torchrun_line.json; echo; tail -5 $O/torchrun.err
