import os
os.environ.setdefault("GPUART_LIBDIR", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpuart_amd", "lib_test"))  # uses test hooks (include/gpuart_hip_test.h)
import sys, os, numpy as np, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
from gpuart_amd import binding as B, synth_scenes as S
W,H=1920,1080
cam=dict(S.BENCH_CAMERA); cam['dir']=S.camera_dir(cam)
be=B.Backend(0)
tree,_=B.compile_bvh(S.scene_d())
be.resize(W,H); be.upload_bvh(tree)
c=B.camera_basis(cam['pos'],cam['dir'],cam['up'],cam['fov_y'],cam['screen_dist'],W,H)
be.set_camera(c)
rs,rd=be.test_cam_rays()
# tile-major order like the renderer: 8x8 tiles
idx=np.arange(W*H).reshape(H,W)
idx=idx.reshape(H//8,8,W//8,8).transpose(0,2,1,3).reshape(-1)
rs=rs.reshape(-1,4)[idx]; rd=rd.reshape(-1,4)[idx]
o0,o1=be.test_traverse(rs,rd,S.USER_SPHERE)
hit=o1[:,3]>=0
P=o0[hit,1:4]; N=o1[hit,:3]
rng=np.random.RandomState(0)
n=len(P)
d=rng.normal(size=(n,3)); d/=np.linalg.norm(d,axis=1,keepdims=True); d*=np.sign((d*N).sum(1,keepdims=True))
rs2=np.zeros((n,4),np.float32); rs2[:,:3]=P; rd2=np.zeros((n,4),np.float32); rd2[:,:3]=d
print("secondary rays",n)
def run(tag,order):
    a=rs2[order]; b=rd2[order]
    be.test_traverse(a,b,S.USER_SPHERE)   # warm
    be.test_traverse(a,b,S.USER_SPHERE)
    print(tag,"done")
run("tile", np.arange(n))
octant=(d[:,0]>0)*1+(d[:,1]>0)*2+(d[:,2]>0)*4
run("octant-stable", np.argsort(octant,kind='stable'))
# octant within blocks of 4096 consecutive rays
blk=np.arange(n)//4096
run("octant-in-4096-blocks", np.lexsort((octant,blk)))
run("random", rng.permutation(n))
