import numpy as np, torch, sys, os
sys.path.insert(0,'.')
import ins_amd as ins
keep=[]
for n in [(32,32,16),(64,16,16),(10,6,16),(64,16,16)]:
    lay = ins.SlabLayout(n, 1, 0)
    K = ins.HipSlabKernels(lay)
    keep.append(K)
    nx,ny,nz = n; kxn = nx//2+1
    rng = np.random.default_rng(3)
    a = rng.standard_normal((nz,ny,nx))
    pI = torch.from_numpy(a.reshape(-1).copy()).cuda()
    work, send = K.cplx(), K.cplx()
    K.fft_forward_xy(pI, work, send)
    torch.cuda.synchronize()
    got = work.cpu().numpy().view(np.complex128).reshape(nz,ny,kxn)
    want = np.fft.rfftn(a, axes=(1,2))
    e1 = np.linalg.norm(got-want)/np.linalg.norm(want)
    gs = send.cpu().numpy().view(np.complex128).reshape(nz,ny,kxn)
    e1b = np.linalg.norm(gs-want)/np.linalg.norm(want)
    # z solve on the (correct) numpy spectrum
    buf = torch.from_numpy(want.reshape(-1).view(np.float64).copy()).cuda()
    K.fft_solve_z(buf)
    torch.cuda.synchronize()
    gz = buf.cpu().numpy().view(np.complex128).reshape(nz,ny,kxn)
    h=[1/v for v in n]; om=np.prod(h)
    sym=lambda a_,cnt: 4*om*np.sin(np.pi*(np.arange(cnt)/n[a_]))**2/h[a_]**2
    den = sym(2,nz)[:,None,None]+sym(1,ny)[None,:,None]+sym(0,kxn)[None,None,:]
    f = np.fft.fft(want, axis=0)
    with np.errstate(divide='ignore',invalid='ignore'): f = -f/den
    f[0,0,0]=0
    wz = np.fft.ifft(f, axis=0)/(nx*ny)
    e2 = np.linalg.norm(gz-wz)/np.linalg.norm(wz)
    # inverse xy on numpy data
    rb = torch.from_numpy(want.reshape(-1).view(np.float64).copy()).cuda()
    out = K.real()
    K.fft_inverse_xy(rb, work, out)
    torch.cuda.synchronize()
    e3 = np.linalg.norm(out.cpu().numpy().reshape(nz,ny,nx)/(nx*ny)-a)/np.linalg.norm(a)
    print(n, "fwd_xy %.2e pack %.2e | solve_z %.2e | inv_xy %.2e" % (e1,e1b,e2,e3), flush=True)
