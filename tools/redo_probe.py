import sys; sys.path.insert(0,'.')
import bench, xee_pkg, numpy as np
xee = xee_pkg.load()
nr,nz,M=1024,512,32
with xee.Ensemble(nr,nz,M) as e:
    e.set_profiles(xee.profiles(nr,nz))
    for m in range(M):
        A,B,Cc,bc,Q = bench.member_fields(nr,nz,m)
        for n,a in (("A",A),("B",B),("C",Cc),("BC_INIT",bc)): e.upload(n,a,first=m)
    e.forcing_from_B(); e.assemble(0)
    for it in (5000, 20000, 50000, 100000):
        e.solve(it,1e-8,-1.0,1.0); print(it, e.timing())
