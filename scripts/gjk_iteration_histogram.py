import sys, numpy as np
sys.path[:0]=['/root/repo','/root/repo/tests']
from constraint_solver_amd import capi
for name,kind,pitch in (("mixed pile",capi.SCENE_MIXED_DROP,1.4),("boxes pile",capi.SCENE_BOXES_DROP,1.8)):
    n=65536
    b,sid=capi.scene_pile(kind,1,n,pitch,4)
    with capi.World(mode=capi.MODE_CONTACTS) as w:
        w.set_polytopes(capi.scene_polytopes(kind)); w.upload(b,sid)
        for _ in range(180): w.step(1/60,20)
        off,nb=w.neighbours(1/60)
        i=np.repeat(np.arange(n),np.diff(off)); keep=nb>i
        pairs=np.stack([i[keep],nb[keep]],axis=1).astype(np.uint32)
        r=w.narrowphase_gjk(pairs)
    print(name,"pairs",len(pairs),"status counts",np.bincount(r["status"],minlength=3))
    for st,nm in ((0,"separated"),(1,"penetrating"),(2,"degenerate")):
        m=r["status"]==st
        if m.any():
            print("  ",nm,"gjk iters: mean %.1f p50 %d p90 %d p99 %d max %d"%(r["gjk_iterations"][m].mean(),*np.percentile(r["gjk_iterations"][m],[50,90,99]).astype(int),r["gjk_iterations"][m].max()), " epa iters: mean %.1f p90 %d max %d"%(r["epa_iterations"][m].mean(),np.percentile(r["epa_iterations"][m],90),r["epa_iterations"][m].max()))
    pen=r["status"]==1
    print("   penetrating depth: p50 %.2e p90 %.2e max %.2e"%tuple(np.percentile(r["depth"][pen],[50,90,100])))
