#!/bin/bash
cd "$GRAFT_REPO_ROOT"
ORBX_BENCH_FORCE_COMM=1 python bench.py --gpus 1 --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-pipelined --no-host-api --no-extra-configs 2>&1 | tail -3 | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        j = json.loads(ln); print(json.dumps(j.get('forced_comm'), indent=1)); print(j.get('gathered_on_rank0'), j['ms_per_step'])
"
python - <<'PY'
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29555")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
a = torch.arange(1000, dtype=torch.uint8, device="cuda") ; b = torch.zeros(1000, dtype=torch.uint8, device="cuda")
w = dist.batch_isend_irecv([dist.P2POp(dist.isend, a, 0), dist.P2POp(dist.irecv, b, 0)])
for x in w: x.wait()
torch.cuda.synchronize()
print("self send/recv equal:", bool((a == b).all()))
v = torch.zeros(64, dtype=torch.float32, device="cuda").view(torch.uint8)
c = torch.zeros(40, dtype=torch.int32, device="cuda"); c[5] = 77
w = dist.batch_isend_irecv([dist.P2POp(dist.isend, c[5:6], 0), dist.P2POp(dist.irecv, c[0:1], 0)])
for x in w: x.wait()
torch.cuda.synchronize(); print("int32 slice:", c[:8].tolist())
dist.destroy_process_group()
PY
