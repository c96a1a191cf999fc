"""Not a test (CPU only): leaves a 64-query group has to scan with the box test alone and with box + slab (bvh.hip leaf slabs), at the optimum and off it.
    python tools/slab_sim.py dragon"""
import sys, numpy as np
import os; REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tools'))
import fgoicp_amd as fg
from scipy.spatial import cKDTree
import kd_sim as K

def slabs(p, perm, leaf=32):
    n=len(p); nleaf=(n+leaf-1)//leaf; pad=nleaf*leaf-n
    q=p[perm]
    if pad: q=np.concatenate([q,np.repeat(q[-1:],pad,0)])
    q=q.reshape(nleaf,leaf,3)
    c=q.mean(1,keepdims=True); d=q-c
    cov=np.einsum('lki,lkj->lij',d,d)
    w,v=np.linalg.eigh(cov)
    nrm=v[:,:,0]  # smallest eigenvalue
    proj=np.einsum('lki,li->lk',q,nrm)
    return nrm, proj.min(1), proj.max(1)

def run(name, angle):
    tgt, src, R_gt, t_gt = fg.synth.workload(name, angle_deg=150.0, min_angle_deg=110.0)
    tgt=tgt.astype(np.float64); q=src.astype(np.float64)@R_gt.T+t_gt
    # perturb: rotate about centroid by `angle` degrees
    rng=np.random.default_rng(1); Rp=fg.synth.random_rotation(rng, angle, angle)
    c=q.mean(0); q=(q-c)@Rp.T+c
    qperm=K.kd_order(q,64); q=q[qperm]
    d,_=cKDTree(tgt).query(q); b2=d*d*(1+1e-5)
    perm=K.kd_order(tgt); lo,hi=K.leaf_boxes(tgt,perm); nrm,a,b=slabs(tgt,perm)
    G=len(q)//64; gs=rng.choice(G,min(200,G),replace=False)
    sc=sc2=0
    for g in gs:
        Q=q[g*64:(g+1)*64]; B=b2[g*64:(g+1)*64]
        wl,wh=Q.min(0),Q.max(0); r2=B.max()
        dd=np.maximum(np.maximum(lo-wh,wl-hi),0); cand=(dd*dd).sum(-1)<=r2
        D=K.box_d2(lo[cand],hi[cand],Q)
        pq=Q@nrm[cand].T  # (64, L)
        S=np.maximum(np.maximum(a[cand][None]-pq, pq-b[cand][None]),0)**2
        sc+=(D<=B[:,None]).any(0).sum()
        sc2+=((D<=B[:,None])&(S<=B[:,None])).any(0).sum()
    print(f"{name} perturbed {angle} deg: mean NN dist {d.mean():.4f}; leaves scanned per 64-query group: box test {sc/len(gs):.1f}, box+slab test {sc2/len(gs):.1f}")
for nm in sys.argv[1:]:
    for ang in (20.0, 60.0, 150.0): run(nm, ang)
