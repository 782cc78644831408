import sys, os, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import conftest
import my_slam_amd as M, my_slam_amd.synth as synth
W,H,n,B=1241,376,2000,32
fr_np=synth.stream(5,W,H,B)
for pitch in (1241, 1244, 1280):
    buf=torch.zeros((B,H,pitch),dtype=torch.uint8,device="cuda")
    buf[:,:,:W]=torch.from_numpy(fr_np).cuda()
    fr=buf[:,:,:W]
    e=M.ORBextractor(n,max_width=W,max_height=H,max_batch=B); cap=e.cap
    k=torch.zeros((B,cap,7),device="cuda"); d=torch.zeros((B,cap,32),dtype=torch.uint8,device="cuda")
    c=torch.zeros(B,dtype=torch.int32,device="cuda"); s=torch.zeros(B,dtype=torch.int32,device="cuda")
    st=torch.cuda.Stream()
    e.set_profiling(True); acc=np.zeros(4)
    for _ in range(8):
        e.extract_batch_device(fr.data_ptr(),B,W,H,fr.stride(1),fr.stride(0),k.data_ptr(),d.data_ptr(),c.data_ptr(),s.data_ptr(),st.cuda_stream)
        acc+=e.stage_ms()
    print("pitch",pitch,"stage_ms",(acc/8).round(4).tolist(),"kp",int(c.sum()))
