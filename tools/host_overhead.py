import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch, ctypes as C
from ivae_amd import ops, _lib
from time_generic import CFG
dev = torch.device("cuda:0")
shape, B, ck, N = CFG["cfg1g"]
g = torch.Generator(device=dev).manual_seed(0)
P = shape.n_params()
params = (torch.rand(P, generator=g, device=dev) - .5) * .2
m = torch.zeros_like(params); v = torch.zeros_like(params)
x = torch.randn(B, shape.data_dim, generator=g, device=dev)
u = torch.zeros(B, shape.aux_dim, device=dev); u[:, 0] = 1
eps = torch.randn(B, shape.latent_dim, generator=g, device=dev)
coef = ops.make_coef(N=N, global_batch=B, **ck)
ws = ops.Workspace()
def t(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n
print("elbo_fwd_bwd host/total us:", t(lambda: ops.elbo_fwd_bwd(shape, coef, params, x, u, eps, ws)))
loss, z, grads = ops.elbo_fwd_bwd(shape, coef, params, x, u, eps, ws)
print("adam host/total us:", t(lambda: ops.adam_step(params, m, v, grads, 1e-3, 1)))
print("torch.empty x3:", t(lambda: (torch.empty(B, 10, device=dev), torch.empty(1, device=dev), torch.empty_like(params))))
dims = shape.dims(B, 1)
buf = ws.get(dims, dev, False)
L = _lib.lib()
zz = torch.empty(B, 10, device=dev); ll = torch.empty(1, device=dev); gg = torch.empty_like(params)
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
def raw():
    L.ivae_elbo_fwd_bwd(C.byref(dims), C.byref(coef), C.c_void_p(params.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(u.data_ptr()), C.c_void_p(eps.data_ptr()), C.c_void_p(zz.data_ptr()), C.c_void_p(ll.data_ptr()), C.c_void_p(gg.data_ptr()), C.c_void_p(buf.data_ptr()), buf.numel() * 4, st)
print("raw ctypes ivae_elbo_fwd_bwd:", t(raw))
print("ws.get:", t(lambda: ws.get(dims, dev, False)))
def ctx():
    with torch.cuda.device(dev): pass
print("cuda.device ctx:", t(ctx))
print("_stream:", t(lambda: torch.cuda.current_stream(dev).cuda_stream))
