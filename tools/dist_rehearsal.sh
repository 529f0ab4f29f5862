#!/bin/bash
# the driver's launch line, 1 rank (real nccl) and 2 / 4 ranks folded onto the one GPU (gloo rehearsal; rank 0 also drives
# the product_merge sections -- comm.hip with folded ranks -- while the other ranks wait on the store)
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --cpu-seconds 0 --no-extras 2>gpurun_out/dist1.err | tail -1 | cut -c1-600
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 5 --warmup 2 --dist-backend gloo --fold-ranks --scale 0.25 --queen-rows 60000 --merge-steps 2 2>gpurun_out/dist2.err | tail -1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 4 --steps 3 --warmup 1 --dist-backend gloo --fold-ranks --scale 0.1 --queen-rows 60000 --merge-steps 2 2>gpurun_out/dist4.err | tail -1
tail -n 3 gpurun_out/dist2.err gpurun_out/dist4.err
