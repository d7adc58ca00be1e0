# Rehearsal of the multi-rank bench flow on ONE GPU: N ranks share the device, collectives over gloo (RCCL refuses two
# ranks on one device).  Checks the code path, not the speed.  Usage: bash scripts/rehearse_sharded.sh [ranks=2] [bench args]
n=${1:-2}; shift
export ISEHR_DIST_BACKEND=gloo ISEHR_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus $n --steps 6 --warmup 2 --no-cpu-baseline --rows 200000 "$@" 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo\]" | tail -3 | cut -c1-700
