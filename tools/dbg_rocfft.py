import numpy as np, torch
torch.manual_seed(0)
def check(shape):
    a = torch.randn(shape, dtype=torch.float64, device="cuda")
    got = torch.fft.rfftn(a, dim=(-2, -1)).cpu().numpy()
    want = np.fft.rfftn(a.cpu().numpy(), axes=(-2, -1))
    print(shape, "rfft2 relerr %.3e" % (np.linalg.norm(got - want) / np.linalg.norm(want)), flush=True)
for shp in [(16, 32, 64), (32, 32, 64), (16, 32, 32), (16, 16, 64), (16, 16, 64), (16, 6, 10), (16, 16, 64)]:
    check(shp)
