import sys, numpy as np, torch
sys.path.insert(0,'.')
import ins_amd as ins
from ins_amd import _lib
from oracle import ins_oracle as o
from tests import fixtures as fx
lib=_lib.load()
def rell2(a,b): return float(np.sqrt(np.sum((a-b)**2))/np.sqrt(np.sum(b**2)))
D,P=ins.DirichletBC,ins.PeriodicBC
for n in (16, 64, 256):
    x=(ins.cosine_grid(0.0,1.0,n),ins.cosine_grid(0.0,1.0,n),np.linspace(-0.2,0.2,33))
    sp=ins.Setup(x=x,Re=100.0,boundary_conditions=((D(),D()),(D(),D()),(P(),P())))
    ps=ins.psolver_direct(sp)
    print('n',n,'fold mask',lib.ins_dbg_fdm_fold_mask(ps.handle))
    if n<=64:
        with _lib.options(INS_DISABLE_FDM_FOLD=1):
            ps0=ins.psolver_direct(sp)
        print('  nofold mask',lib.ins_dbg_fdm_fold_mask(ps0.handle))
        f=ins.scalarfield(sp); f.copy_(torch.randn(f.shape,dtype=torch.float64,device=f.device))
        a=ins.to_numpy(ins.poisson(ps,f)); b=ins.to_numpy(ins.poisson(ps0,f))
        ip=tuple(slice(lo,hi) for lo,hi in sp.grid.Ip)
        print('  fold vs nofold rel', rell2(a[ip],b[ip]))
# symmetric eigenvector check on host for n=256
import ins_amd.pressure as pr
sp=ins.Setup(x=(ins.cosine_grid(0.0,1.0,256),)*2+(np.linspace(-0.2,0.2,33),),Re=100.0,boundary_conditions=((D(),D()),(D(),D()),(P(),P())))
ps=ins.psolver_direct(sp)
V=ps._V[0]
worst=0
for j in range(V.shape[1]):
    v=V[:,j]; se=np.sum((v-v[::-1])**2); so=np.sum((v+v[::-1])**2); nn=np.sum(v*v)
    worst=max(worst,min(se,so)/nn)
print('worst asymmetry (rel^2)',worst)
