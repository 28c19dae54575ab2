import os, sys, time, numpy as np
sys.path.insert(0,'/root/repo')
import pwnfps_amd
GOLD='/root/repo/tests/golden'
sph=np.load(GOLD+'/spheres_t0.npy')
for n in (1,2,4,8):
    r = pwnfps_amd.Renderer(256,128, devices=[0]*n) if n>1 else pwnfps_amd.Renderer(256,128)
    r.level_load(GOLD+'/levels/pwnfps_level.txt'); r.set_objects(sph)
    import torch
    for rep in range(2):
        t0=time.perf_counter()
        for i in range(3000): r.set_objects(sph)
        torch.cuda.synchronize()
        dt=time.perf_counter()-t0
    print("members %d: set_objects %.1f us"%(n, dt/3000*1e6), flush=True)
    r.close()
