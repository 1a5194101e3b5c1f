"""Shared helpers for the parity tests: fixture loading and error metrics."""
import glob
import json
import os

import numpy as np

from oracle.np_ref import Coef, Spec

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def case_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz"))
                  if not os.path.basename(p).startswith(("mcc", "trace_")))


class Case:
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.meta = json.loads(str(self.z["meta"]))
        m = self.meta
        self.spec = Spec(data_dim=m["data_dim"], latent_dim=m["latent_dim"], aux_dim=m["aux_dim"],
                         hidden_dim=m["hidden_dim"], n_layers=m["n_layers"], activation=m["activation"],
                         slope=m["slope"], decoder_var=m["decoder_var"])
        self.family = m["family"]

    def coef(self, step=0):
        c = self.meta["steps"][step]
        return Coef(a=c["a"], b=c["b"], c=c["c"], d=c["d"], N=c["N"], pairwise=c["pairwise"])

    def group(self, prefix):
        return {k[len(prefix):]: self.z[k] for k in self.z.files if k.startswith(prefix)}

    @property
    def params(self):
        return self.group("params.")

    @property
    def grads(self):
        return self.group("grad.")

    def inputs(self, step=0):
        pre = "" if step == 0 else f"step{step}."
        return self.z[pre + "x"], self.z[pre + "u"], self.z[pre + "eps"]

    def loss(self, step=0):
        return float(self.z["loss" if step == 0 else f"step{step}.loss"])

    def has_adam(self, k):
        return any(f.startswith(f"adam{k}.p.") for f in self.z.files)


def relmax(a, b):
    """per-tensor max-norm relative error (SURVEY.md §7 'hard parts')."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
