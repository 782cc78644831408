import sys, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import numpy as np, torch
import conftest, oracle_lib as O
import my_slam_amd as M, my_slam_amd.synth as synth
W,H,n,seed=int(sys.argv[1]),int(sys.argv[2]),int(sys.argv[3]),int(sys.argv[4])
img=synth.texture(seed,W,H)
ex=M.ORBextractor(n,1.2,8,20,7,max_width=W,max_height=H)
k,d=ex(img)
ok,od,_=O.Extractor(n).extract(img)
print(len(k),len(ok))
for l in range(8):
    a=k[k["octave"]==l]; b=ok[ok["octave"]==l]
    same = len(a)==len(b) and np.array_equal(a["x"],b["x"]) and np.array_equal(a["y"],b["y"])
    sa=set(zip(a["x"].tolist(),a["y"].tolist())); sb=set(zip(b["x"].tolist(),b["y"].tolist()))
    print("level",l,"gpu",len(a),"oracle",len(b),"ordered-equal",same,"set-equal",sa==sb,"ncand",len(ex.candidates(0,l)))
