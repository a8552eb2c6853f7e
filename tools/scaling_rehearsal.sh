#!/bin/bash
# One-GPU rehearsal of bench.py's N > 1 paths (the driver owns the real 8-GPU run):
#  1. the default N=1 line
#  2. N=2 launched exactly as the driver does, but with gloo so both ranks share
#     the one GPU (YTHIP_DIST_BACKEND): weak primary + configs2_strong, side-stream gather
#  3. the slice rank 0 of 8 would render in the weak-scaling frame (3584 wide), no gather
# Usage: tools/scaling_rehearsal.sh OUT
out=${1:-gpurun_out/scaling_rehearsal.txt}
mkdir -p $(dirname $out); : > $out
echo "== N=1 default" >> $out
timeout 600 python bench.py >> $out 2>&1
echo "== N=2 rehearsal (gloo, both ranks on one GPU)" >> $out
YTHIP_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 2 --warmup 1 >> $out 2>&1
echo "== N=2 rehearsal, overlapped gather (experiment)" >> $out
YTHIP_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 2 --warmup 1 --overlap-gather --scaling weak >> $out 2>&1
echo "== N=1 with a world_size-1 RCCL group: gather on the kernel stream / overlapped / none" >> $out
for f in "--rehearse-gather" "--rehearse-gather --overlap-gather" ""; do
  timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 5 $f 2>&1 | grep '^{' | cut -c1-140 >> $out
done
for n in 2 4 8; do
  echo "== weak frame, slice of rank 0/$n" >> $out
  res=$(python -c "import bench; print(bench.weak_resolution(1280, $n))")
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --as-rank 0/$n --resolution $res >> $out 2>&1
done
grep -c . $out
