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
for n in names:
    k = P[n].size
    if n == (sys.argv[3] if len(sys.argv) > 3 else "f.fc.1.weight"):
        G = gm[off:off+k].reshape(P[n].shape); R = M[n]
        E = np.abs(G - R) / np.abs(R).max()
        print("tile error map (rows n/16, cols k/16):")
        for i in range(0, E.shape[0], 16):
            print(["%.1e" % E[i:i+16, j:j+16].max() for j in range(0, E.shape[1], 16)])
        bad = np.argwhere(E > 1e-3)
        print("bad count", len(bad), "rows", sorted(set(bad[:,0]))[:70], "cols", sorted(set(bad[:,1]))[:70])
    off += k
gv = v1[0].cpu().numpy(); P0f = flat(c.params, names); off = 0
for n in names:
    k = P[n].size
    if n == (sys.argv[3] if len(sys.argv) > 3 else "f.fc.1.weight"):
        a, b = 1495, 1506
        print("m got", gm[off+a:off+b]); print("m ref", M[n].reshape(-1)[a:b])
        print("v got", gv[off+a:off+b]); print("dp got", (got - P0f)[off+a:off+b]); 
        a, b = 2045, 2056
        print("m got", gm[off+a:off+b]); print("m ref", M[n].reshape(-1)[a:b]); print("dp got", (got - P0f)[off+a:off+b])
    off += k
off = 0
for n in names:
    k = P[n].size
    if n == (sys.argv[3] if len(sys.argv) > 3 else "f.fc.1.weight"):
        e = np.abs(gm[off:off+k] - M[n].reshape(-1)) > 1e-3 * np.abs(M[n]).max()
        idxs = np.nonzero(e)[0]
        # contiguous ranges
        rs = []; st = None
        for i in idxs:
            if st is None: st = pv = i
            elif i != pv + 1: rs.append((st, pv)); st = i
            pv = i
        if st is not None: rs.append((st, pv))
        print("bad ranges", rs[:40], "n", len(idxs))
        ratio = gm[off:off+k][idxs] / M[n].reshape(-1)[idxs]
        print("ratio stats", np.percentile(ratio, [0, 50, 100]))
    off += k
