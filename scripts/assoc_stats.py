import sys, numpy as np
sys.path.insert(0,'.')
import torch
from multiviewstitch_amd import _lib, deformation, srt as srt_mod, scene as S
import bench
dev=torch.device('cuda',0)
sc=S.make_scene(3,device=dev)
d=deformation.Deformation(sc.verts,sc.normals,sc.faces)
K=d.UniformSampling(16)
tp,tn=bench.build_target(torch,srt_mod,S,sc,range(8),dev)
d.set_target_dev(tp.data_ptr(),tn.data_ptr(),tp.shape[0],0)
print('nan normals', int(torch.isnan(tn).any(1).sum()), 'of', tp.shape[0])
for it in range(6):
    st=d.iterate(1)
    nt=d.node_targets()
    c=nt['counts'][:,0]; d2=np.sqrt(nt['d2min'])
    print(it,'valid',st['n_valid'],'ball pct 50/90/99/max',np.percentile(c,[50,90,99,100]),'>=10000:',(c>=10000).sum(),'dmin pct',np.percentile(d2,[50,90,99,100]))
