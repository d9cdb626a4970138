import os
os.environ.setdefault("GPUART_LIBDIR", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpuart_amd", "lib_test"))  # uses test hooks (include/gpuart_hip_test.h)
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gpuart_amd import binding as B
from gpuart_amd import synth_scenes as S
from oracle import oracle as O
f32=np.float32
seed=int(sys.argv[1]) if len(sys.argv)>1 else 42874
case=S.random_wild_case(seed)
prims,W,H=case["prims"],case["W"],case["H"]
cd=case["cam"]
cam=O.camera(cd["pos"],cd["dir"],cd["up"],cd["fov_y"],cd["screen_dist"],W,H)
tree,_=O.build_bvh(prims)
sun=O.sun_direction(case["sun_az"],case["sun_alt"])
be=B.Backend(0)
be.resize(W,H); be.upload_bvh(tree); be.set_camera(cam)
seeds=O.randseeds(case["passes"],seed=5489+seed)
def run(ms,npaths,K,sun_on,flags,mode=2):
    P=O.make_params(sun,case["sun_alt"],sun_on,case["user_sphere"],case["us_em"],flags,float(cam[12]),cam[0:3],ms,0.01)
    acc=np.zeros((H,W,4),f32)
    for k in range(K): O.pt_pass(tree,cam,W,H,P,seeds[k],npaths,acc)
    gp=B.Params(); C.memmove(C.byref(gp),C.byref(P),C.sizeof(gp))
    be.set_mode(mode); be.pt_reset(); be.pt_plan(K)
    for k in range(K): be.pt_pass(gp,seeds[k],npaths)
    got=be.read(1)
    same=(got[...,:3].view(np.uint32)==acc[...,:3].view(np.uint32))|((got[...,:3]==0)&(acc[...,:3]==0))|(np.isnan(got[...,:3])&np.isnan(acc[...,:3]))
    bad=~same.all(-1)
    return int(bad.sum()),got,acc,bad
for ms in (1,2,3):
    for sun_on in (False,True):
        for flags in (0,case["us_flags"]):
            n,got,acc,bad=run(ms,1,1,sun_on,flags)
            print("ms %d sun %d flags %d: %d differ"%(ms,sun_on,flags,n))
n,got,acc,bad=run(1,1,1,True,case["us_flags"])
if n==0: n,got,acc,bad=run(case["max_segments"],1,1,True,case["us_flags"])
ys,xs=np.nonzero(bad)
for y,x in list(zip(ys,xs))[:10]:
    print("pixel",x,y,"gpu",got[y,x,:3],got[y,x,:3].view(np.uint32),"oracle",acc[y,x,:3],acc[y,x,:3].view(np.uint32))
# primary-ray traversal: GPU hook vs oracle
rs,rd=O.cam_rays(cam,W,H)
rs=rs.reshape(-1,4); rd=rd.reshape(-1,4)
us=np.array(case["user_sphere"],f32)
g=be.test_traverse(rs,rd,us)
o=O.traverse(tree,rs,rd,us)
for i,(a,b) in enumerate(zip(g,o)):
    a=np.asarray(a); b=np.asarray(b)
    same=(a.view(np.uint32)==b.view(np.uint32))|(np.isnan(a)&np.isnan(b))
    print("traverse output",i,"rows differing:",int((~same.all(-1)).sum()) if a.ndim>1 else int((~same).sum()))
    if a.ndim>1:
        r=np.nonzero(~same.all(-1))[0][:5]
        for k in r: print("   ray",k,"gpu",a[k],"oracle",b[k])
be.close()

# ---- second segment by hand: bounce off the primary hit, both implementations step by step
be=B.Backend(0); be.resize(W,H); be.upload_bvh(tree); be.set_camera(cam)
o0,o1=o
hit=o1[:,3]>=0
P=o0[hit][:,1:4]; N=o1[hit][:,0:3]
print("primary hits:",int(hit.sum()),"of",len(hit))
def bits_same(a,b):
    a=np.asarray(a,np.float32); b=np.asarray(b,np.float32)
    return (a.view(np.uint32)==b.view(np.uint32))|(np.isnan(a)&np.isnan(b))
for k in range(3):
    ri=(P+seeds[k][:3]).astype(np.float32)   # intersection + RandSeed.xyz
    v4=np.zeros((len(P),4),f32); v4[:,:3]=N
    r4=np.zeros((len(P),4),f32); r4[:,:3]=ri
    dg=be.test_hemisphere(v4,r4); do=O.hemisphere(v4,r4)
    sm=bits_same(dg[:,:3],do[:,:3]).all(-1)
    print("pass",k,"hemisphere directions differing:",int((~sm).sum()))
    for q in np.nonzero(~sm)[0][:5]: print("   N",N[q],"ri",ri[q],"gpu",dg[q],"oracle",do[q])
    rs2=np.zeros((len(P),4),f32); rs2[:,:3]=P
    rd2=np.zeros((len(P),4),f32); rd2[:,:3]=do[:,:3]
    g2=be.test_traverse(rs2,rd2,us); o2=O.traverse(tree,rs2,rd2,us)
    for i2,(a,b) in enumerate(zip(g2,o2)):
        sm=bits_same(a,b).all(-1)
        print("   bounce traverse output",i2,"rows differing:",int((~sm).sum()))
        for q in np.nonzero(~sm)[0][:6]: print("      ray",q,"o",rs2[q,:3],"d",rd2[q,:3],"gpu",a[q],"oracle",b[q])
    # and the Sun-shadow query from the hit point
    rd3=np.zeros((len(P),4),f32); rd3[:,:3]=sun
    g3=be.test_traverse(rs2,rd3,us); o3=O.traverse(tree,rs2,rd3,us)
    sm=bits_same(g3[1][:,3],o3[1][:,3])
    print("   sun query type differing:",int((~sm).sum()))
be.close()
