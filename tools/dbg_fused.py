"""Per-tensor error report of ivae_train_steps against the CPU oracle (debug aid for kernel work).
usage: dbg_fused.py [fixture | shape:dd,dl,M] [n_steps] [tensor-name for a 16x16 tile error map]"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from ivae_amd import ops
from oracle import np_ref
from tests.util import Case
from tests.test_gpu_fused import oracle_run, make_problem, flat
name = sys.argv[1] if len(sys.argv) > 1 else "vae_plain_cfg2"
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
if name.startswith("shape:"):
    dd, dl, M = map(int, name[6:].split(","))
    from oracle import torch_ref
    spec = np_ref.Spec(dd, dl, M)
    class _C: pass
    c = _C(); c.params = {k: v.numpy() for k, v in torch_ref.xavier_params(spec, 0).items()}
else:
    c = Case(name); spec = c.spec
names = spec.param_names()
shape = ops.ModelShape(spec.data_dim, spec.latent_dim, spec.aux_dim, spec.hidden_dim, 3, "lrelu", .1)
X, U, idx, eps = make_problem(spec, 1, n_steps)
d = torch.device("cuda:0"); t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(d)
params = t(flat(c.params, names)[None]); m1 = torch.zeros_like(params); v1 = torch.zeros_like(params)
coef = ops.make_coef(decoder_var=spec.decoder_var, global_batch=64)
loss, z = ops.train_steps(shape, coef, params, m1, v1, t(X), t(U) if U is not None else None, t(idx), 0, 1e-2, eps=t(eps), want_z=True)
P, M, V, losses, zl = oracle_run(spec, c.params, X[0], U[0] if U is not None else None, idx[0], eps[0], 1e-2, np_ref.Coef())
print("loss", loss.cpu().numpy(), losses)
got = params[0].cpu().numpy(); gm = m1[0].cpu().numpy(); off = 0
for n in names:
    k = P[n].size
    e = np.abs(got[off:off+k] - P[n].reshape(-1)); em = np.abs(gm[off:off+k]-M[n].reshape(-1)).max()/ (np.abs(M[n]).max()+1e-30)
    print(f"{n:18s} perr max {e.max():.2e} mean {e.mean():.2e}  m relerr {em:.2e}  argmax {e.argmax()} shape {P[n].shape}")
    off += k
off = 0
if len(sys.argv) > 3:
    off = 0
    for n in names:
        k = P[n].size
        if n == sys.argv[3]:
            G = gm[off:off + k].reshape(P[n].shape)
            E = np.abs(G - M[n]) / np.abs(M[n]).max()
            print("exp_avg tile error map of", n, "(rows /16, cols /16):")
            for i in range(0, E.shape[0], 16):
                print(["%.1e" % E[i:i + 16, j:j + 16].max() for j in range(0, max(E.shape[1] if E.ndim > 1 else 1, 1), 16)]
                      if E.ndim > 1 else "%.1e" % E[i:i + 16].max())
        off += k
