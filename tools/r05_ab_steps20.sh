#!/bin/bash
# the driver's command (--steps 20 --warmup 5) against 200 steps, and what the host's way of waiting
# for the GPU (interrupt or polling: HSA_ENABLE_INTERRUPT) does to the short region's edges
F="--no-align --no-extra --no-cpu-baseline"
show='import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(b["value"]/1e6,2), "M/s  wall", round(b["ms_per_step"]*1e3,2), "us  events", round(b["roofline"]["kernel_ms"]*1e3,2), "us")'
for rep in 1 2 3; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 $F 2>/dev/null | python3 -c "$show" "steps20 default      "
  HSA_ENABLE_INTERRUPT=0 python3 bench.py --gpus 1 --steps 20 --warmup 5 $F 2>/dev/null | python3 -c "$show" "steps20 polling      "
  python3 bench.py --gpus 1 --steps 200 --warmup 20 $F 2>/dev/null | python3 -c "$show" "steps200 default     "
  HSA_ENABLE_INTERRUPT=0 python3 bench.py --gpus 1 --steps 200 --warmup 20 $F 2>/dev/null | python3 -c "$show" "steps200 polling     "
done
