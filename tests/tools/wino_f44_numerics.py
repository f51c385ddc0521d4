"""fp32 error of Winograd F(4x4, 3x3) + direct z (the next candidate form of the k3 s1 layers, DESIGN work list) beside the
F(2x2, 3x3) form the kernels use, against a float64 direct convolution (numpy, CPU): rel. L2 ~2e-6 vs ~3e-7."""
import numpy as np
rng=np.random.default_rng(0)
def mats(m):
    if m==2:
        BT=np.array([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]],float)
        G=np.array([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]],float)
        AT=np.array([[1,1,1,0],[0,1,-1,-1]],float)
    else:
        BT=np.array([[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]],float)
        G=np.array([[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]],float)
        AT=np.array([[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]],float)
    return BT,G,AT
def run(m,CI,CO,T,dt):
    BT,G,AT=mats(m); a=m+2
    d=rng.standard_normal((3,T,CI,a,a))          # 3 z planes, T tiles
    g=rng.standard_normal((3,CI,CO,3,3))*0.1
    # reference fp64 direct
    ref=np.zeros((T,CO,m,m))
    for kz in range(3):
        for y in range(m):
            for x in range(m):
                ref[:,:,y,x]+=np.einsum('tcij,coij->to',d[kz][:,:,y:y+3,x:x+3],g[kz])
    d32=d.astype(dt); g32=g.astype(dt); BTf,Gf,ATf=BT.astype(dt),G.astype(dt),AT.astype(dt)
    U=np.einsum('ai,zcoij,bj->zcoab',Gf,g32,Gf).astype(dt)
    V=np.einsum('ai,ztcij,bj->ztcab',BTf,d32,BTf).astype(dt)
    M=np.zeros((T,CO,a,a),dt)
    for kz in range(3):
        for c in range(CI):                       # sequential fp32 accumulation like the MFMA chain
            M+= (V[kz][:,c,None,:,:]*U[kz][None,c,:,:,:]).astype(dt)
    Y=np.einsum('ya,toab,xb->toyx',ATf,M,ATf).astype(dt)
    e=Y.astype(float)-ref
    return np.sqrt((e**2).sum()/(ref**2).sum()), np.abs(e).max()/np.abs(ref).max()
for CI,CO in ((16,16),(32,32),(8,8)):
    for m in (2,4):
        r=run(m,CI,CO,256,np.float32)
        print(f"F({m}x{m},3x3) CI={CI} CO={CO}: rel L2 {r[0]:.2e}  max|err|/max|ref| {r[1]:.2e}")
