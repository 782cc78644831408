#!/bin/bash
# timing ablation of k_fast_cells (ORBX_DBG_STAGE: 1 tile load only, 2 +stage 1, 3 +stage 2, 0 full)
for st in ${STAGES:-1 2 3 0}; do
  ORBX_DBG_STAGE=$st python - <<PY
import sys, os, numpy as np, torch
sys.path.insert(0,'tests'); import conftest
import my_slam_amd as M, my_slam_amd.synth as synth
B=64; frames=torch.from_numpy(synth.stream(4,640,480,B)).cuda()
ex=M.ORBextractor(1000,max_width=640,max_height=480,max_batch=B); cap=ex.cap
k=torch.zeros((B,cap,7),device='cuda'); d=torch.zeros((B,cap,32),dtype=torch.uint8,device='cuda'); c=torch.zeros(B,dtype=torch.int32,device='cuda'); s=torch.zeros(B,dtype=torch.int32,device='cuda')
st=torch.cuda.Stream(); torch.cuda.set_stream(st)
ex.set_profiling(True); acc=np.zeros(4)
for i in range(12):
    ex.extract_batch_device(frames.data_ptr(),B,640,480,frames.stride(1),frames.stride(0),k.data_ptr(),d.data_ptr(),c.data_ptr(),s.data_ptr(),st.cuda_stream)
    if i>=2: acc+=ex.stage_ms()
print("dbg",os.environ.get("ORBX_DBG_STAGE"),"stage ms",(acc/10).round(4))
PY
done
