"""FCNet with the reference's constructor and checkpoint format, backed by the HIP engine.

Mirrors NSFnet/net.py:22-54 (identical in ev-NSFnet/): ``FCNet(num_ins, num_outs,
num_layers, hidden_size, activation)`` where ``num_layers`` counts HIDDEN layers, Tanh
follows every Linear but the last, and ``state_dict()`` keys are
``layers.layer_{i}.weight|bias``.  The parameters live on the device as ONE flat fp32
vector (engine.DeviceNet); ``forward`` runs the HIP value-mode kernel.
"""
from collections import OrderedDict

import numpy as np
import torch

from . import engine as _eng


def default_init_flat(num_ins, num_outs, num_layers, hidden_size):
    """torch's default nn.Linear initialisation, consuming the global torch RNG exactly as the
    reference FCNet constructor does (one Linear after another, net.py:38-46), so a caller
    that sets torch.manual_seed gets the reference's initial weights."""
    widths = [num_ins] + [hidden_size] * num_layers + [num_outs]
    parts = []
    for i in range(len(widths) - 1):
        lin = torch.nn.Linear(widths[i], widths[i + 1])
        parts += [lin.weight.detach().reshape(-1), lin.bias.detach().reshape(-1)]
    return torch.cat(parts)


class FCNet:
    def __init__(self, num_ins=3, num_outs=3, num_layers=10, hidden_size=50, activation=torch.nn.Tanh,
                 device=None):
        if num_ins != 2:
            raise ValueError("the MI355X engine implements the steady 2D problem: num_ins must be 2")
        if activation not in (torch.nn.Tanh, None):
            raise ValueError("only tanh activations are implemented (the reference never uses another)")
        self.num_ins, self.num_outs, self.num_layers, self.hidden_size = num_ins, num_outs, num_layers, hidden_size
        self.depth = num_layers + 1
        init = default_init_flat(num_ins, num_outs, num_layers, hidden_size)   # consumes torch RNG first
        dev = torch.device(device if device is not None else "cuda:0")
        self.dev_net = _eng.DeviceNet(num_outs, num_layers, hidden_size, dev)
        self.dev_net.set_flat(init)

    # ---- torch.nn.Module-like surface used by the reference scripts ----
    def state_dict(self):
        return OrderedDict(self.dev_net.state_dict())

    def load_state_dict(self, sd, strict=True):
        sd = {k[len("module."):] if k.startswith("module.") else k: v for k, v in sd.items()}
        self.dev_net.load_state_dict(sd)

    def parameters(self):
        """Views of the flat device parameter vector, in state_dict order."""
        off, out = 0, []
        for _, shape in self.dev_net.keys_and_shapes():
            n = int(np.prod(shape))
            out.append(self.dev_net.params[off:off + n].view(shape))
            off += n
        return out

    def to(self, device):
        return self

    def eval(self):
        return self

    def train(self, mode=True):
        return self

    def forward(self, X):
        X = torch.as_tensor(X)
        x = X[:, 0].detach().float().cpu().numpy()
        y = X[:, 1].detach().float().cpu().numpy()
        plan = _eng.ValuePlan(self.dev_net, x, y, with_backward=False)
        plan.forward(save=False)
        return plan.pred.t().contiguous()

    __call__ = forward
