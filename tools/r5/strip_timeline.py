import os, sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import pwnfps_amd
GOLD='/root/repo/tests/golden'
w,h=3840,2160
r=pwnfps_amd.Renderer(w,h)
r.level_load(GOLD+'/levels/pwnfps_level.txt'); r.set_objects(np.load(GOLD+'/spheres_t0.npy'))
_,_,spawn=r.get_level(); cam=pwnfps_amd.spawn_camera(spawn)
sb=np.zeros((h,w),np.uint32); r.host_register(sb)
for k in (-1, 8):
    r.set_call_strips(k)
    for i in range(3): r.trace_screen_centred(cam,0.0,want_z=False,sbuf=sb)
    os.environ['PWN_DBG_STRIP_TIMELINE']='1'
    for i in range(2): r.trace_screen_centred(cam,0.0,want_z=False,sbuf=sb)
    del os.environ['PWN_DBG_STRIP_TIMELINE']
