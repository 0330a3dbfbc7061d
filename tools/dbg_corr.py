import numpy as np, torch, sys, os
sys.path.insert(0,'.')
import ins_amd as ins
out = sys.argv[1]
res = {}
for n in [(16,12,20),(64,8,8),(8,64,8),(8,8,64),(130,16,16)]:
    x = tuple(np.linspace(0.0, 1.0, ni + 1) for ni in n)
    sp = ins.Setup(x=x, Re=500.0)
    ps = ins.psolver_spectral(sp)
    u = ins.random_field(sp, kp=2, seed=3, psolver=ps)
    m = ins.RKMethods.RK44(); cache = ins.ode_method_cache(m, sp, ps)
    st = ins.create_stepper(m, setup=sp, psolver=ps, u=u, t=0.0)
    st = ins.timestep_(m, st, 0.01, cache=cache)
    res[str(n)] = ins.to_numpy(st.u)
    del ps, sp, cache, st, u
    import gc; gc.collect()
np.savez(out, **res)
