import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import conftest, oracle_lib as O
import my_slam_amd as M, my_slam_amd.synth as synth
img=synth.texture(1,640,480)
ex=M.ORBextractor(1000,max_width=640,max_height=480)
k,d=ex(img); ok,od,_=O.Extractor(1000).extract(img)
print(len(k),len(ok), (k.tobytes()==ok.tobytes()))
bad=np.nonzero((d!=od).any(axis=1))[0]
print('bad rows',len(bad))
for i in bad[:20]:
    nb=int(np.unpackbits(d[i]^od[i]).sum())
    print(i, ok[i]['x'],ok[i]['y'],ok[i]['octave'],ok[i]['angle'], 'bits differ',nb)
