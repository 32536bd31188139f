"""Forward-mode restatement of the PINN step in numpy (TEST INFRASTRUCTURE).

The HIP kernels do not run nine reverse sweeps; they propagate four streams
per activation - value a, a_x, a_y and the Laplacian a_D = a_xx + a_yy - in
ONE forward sweep and then reverse-differentiate that sweep by hand.  This
module is the executable specification of that algorithm (fp64 by default),
checked against the autograd restatement and the reference-generated golden
fixtures in tests/.  Only Laplacians are needed because the reference only
consumes u_xx+u_yy and v_xx+v_yy (NSFnet/pinn_solver.py:159-160,
ev-NSFnet/pinn_solver.py:337-338).

Parameter layout: flat vector in state_dict order
``layer_0.weight (H,2) | layer_0.bias (H) | ... | layer_L.weight (n_out,H) |
layer_L.bias (n_out)`` (NSFnet/net.py:36-46).
"""
import numpy as np


def layer_shapes(n_in, n_out, n_hidden, hidden):
    widths = [n_in] + [hidden] * n_hidden + [n_out]
    return [(widths[i + 1], widths[i]) for i in range(len(widths) - 1)]


def param_count(n_in, n_out, n_hidden, hidden):
    return sum(o * i + o for o, i in layer_shapes(n_in, n_out, n_hidden, hidden))


def unflatten(flat, n_in, n_out, n_hidden, hidden):
    flat = np.asarray(flat)
    out, off = [], 0
    for o, i in layer_shapes(n_in, n_out, n_hidden, hidden):
        W = flat[off:off + o * i].reshape(o, i); off += o * i
        b = flat[off:off + o]; off += o
        out.append((W, b))
    assert off == flat.size
    return out


def flatten(pairs):
    return np.concatenate([np.concatenate([W.reshape(-1), b.reshape(-1)]) for W, b in pairs])


# --------------------------------------------------------------------------
# 4-stream forward (value, d/dx, d/dy, Laplacian) of net(x, y)
# --------------------------------------------------------------------------
def forward4(params, x, y):
    """Returns out (N, n_out, 4) and per-hidden-layer saved tuples
    (t, z_x, z_y, z_D) each (N, H) - exactly what the HIP forward saves."""
    x = np.asarray(x).reshape(-1); y = np.asarray(y).reshape(-1)
    W0, b0 = params[0]
    z = np.outer(x, W0[:, 0]) + np.outer(y, W0[:, 1]) + b0
    zx = np.broadcast_to(W0[:, 0], z.shape).copy()
    zy = np.broadcast_to(W0[:, 1], z.shape).copy()
    zd = np.zeros_like(z)
    saved = []
    n_lin = len(params)
    for l in range(n_lin - 1):
        t = np.tanh(z)
        d1 = 1.0 - t * t
        d2 = -2.0 * t * d1
        saved.append((t, zx, zy, zd))
        a, ax, ay = t, d1 * zx, d1 * zy
        ad = d2 * (zx * zx + zy * zy) + d1 * zd
        W, b = params[l + 1]
        z, zx, zy, zd = a @ W.T + b, ax @ W.T, ay @ W.T, ad @ W.T
    out = np.stack([z, zx, zy, zd], axis=2)  # (N, n_out, 4)
    return out, saved


def backward4(params, x, y, saved, out_adj):
    """Reverse pass of forward4.  out_adj (N, n_out, 4) = dL/d out.
    Returns flat parameter gradient and per-layer z-adjoints (for kernel tests)."""
    x = np.asarray(x).reshape(-1); y = np.asarray(y).reshape(-1)
    n_lin = len(params)
    grads = [None] * n_lin
    zbar_all = [None] * (n_lin - 1)
    # output layer
    W, b = params[-1]
    t, zx, zy, zd = saved[-1]
    d1 = 1.0 - t * t; d2 = -2.0 * t * d1
    a_streams = (t, d1 * zx, d1 * zy, d2 * (zx * zx + zy * zy) + d1 * zd)
    gW = sum(out_adj[:, :, s].T @ a_streams[s] for s in range(4))
    gb = out_adj[:, :, 0].sum(axis=0)
    grads[-1] = (gW, gb)
    g = [out_adj[:, :, s] @ W for s in range(4)]  # adjoints of a-streams of last hidden layer
    for l in range(n_lin - 2, -1, -1):
        t, zx, zy, zd = saved[l]
        d1 = 1.0 - t * t; d2 = -2.0 * t * d1; d3 = -2.0 * d1 * (1.0 - 3.0 * t * t)
        ga, gx, gy, gd = g
        zb_x = d1 * gx + 2.0 * d2 * zx * gd
        zb_y = d1 * gy + 2.0 * d2 * zy * gd
        zb_d = d1 * gd
        zb = d1 * ga + d2 * (zx * gx + zy * gy) + (d3 * (zx * zx + zy * zy) + d2 * zd) * gd
        zbar_all[l] = (zb, zb_x, zb_y, zb_d)
        if l == 0:
            gW0 = np.stack([x @ zb + zb_x.sum(axis=0), y @ zb + zb_y.sum(axis=0)], axis=1)
            grads[0] = (gW0, zb.sum(axis=0))
        else:
            tp, zxp, zyp, zdp = saved[l - 1]
            d1p = 1.0 - tp * tp; d2p = -2.0 * tp * d1p
            ap = (tp, d1p * zxp, d1p * zyp, d2p * (zxp * zxp + zyp * zyp) + d1p * zdp)
            zbs = (zb, zb_x, zb_y, zb_d)
            gW = sum(zbs[s].T @ ap[s] for s in range(4))
            grads[l] = (gW, zb.sum(axis=0))
            Wl = params[l][0]
            g = [zbs[s] @ Wl for s in range(4)]
    return flatten(grads), zbar_all


# --------------------------------------------------------------------------
# value-only forward / backward (BC points, supervised points, entropy net)
# --------------------------------------------------------------------------
def forward1(params, x, y):
    X = np.stack([np.asarray(x).reshape(-1), np.asarray(y).reshape(-1)], axis=1)
    a, saved = X, []
    for l, (W, b) in enumerate(params):
        z = a @ W.T + b
        if l < len(params) - 1:
            a = np.tanh(z); saved.append(a)
        else:
            a = z
    return a, saved


def backward1(params, x, y, saved, out_adj):
    X = np.stack([np.asarray(x).reshape(-1), np.asarray(y).reshape(-1)], axis=1)
    n_lin = len(params)
    grads = [None] * n_lin
    g = out_adj
    for l in range(n_lin - 1, -1, -1):
        a_prev = X if l == 0 else saved[l - 1]
        if l < n_lin - 1:
            t = saved[l]
            g = g * (1.0 - t * t)
        grads[l] = (g.T @ a_prev, g.sum(axis=0))
        g = g @ params[l][0]
    return flatten(grads)


# --------------------------------------------------------------------------
# residuals, loss and full gradient
# --------------------------------------------------------------------------
def residuals(out, Re, vis_t=None, e=None, scale=1.0):
    """eq1..eq3 (and eq4 if e is given) from the 4-stream outputs.
    NSFnet/pinn_solver.py:159-161; ev-NSFnet/pinn_solver.py:311-341."""
    u, v = out[:, 0, 0], out[:, 1, 0]
    s, s2 = scale, scale * scale
    u_x, u_y, lap_u = out[:, 0, 1] * s, out[:, 0, 2] * s, out[:, 0, 3] * s2
    v_x, v_y, lap_v = out[:, 1, 1] * s, out[:, 1, 2] * s, out[:, 1, 3] * s2
    p_x, p_y = out[:, 2, 1] * s, out[:, 2, 2] * s
    nu = 1.0 / Re + (0.0 if vis_t is None else np.asarray(vis_t).reshape(-1))
    eq1 = (u * u_x + v * u_y) + p_x - nu * lap_u
    eq2 = (u * v_x + v * v_y) + p_y - nu * lap_v
    eq3 = u_x + v_y
    eqs = [eq1, eq2, eq3]
    if e is not None:
        eqs.append(eq1 * (u - 0.5) + eq2 * (v - 0.5) - np.asarray(e).reshape(-1))
    return eqs


def pde_loss_and_grad(params, x, y, Re, alpha_e=1.0, vis_t=None, e=None, w=None,
                      scale=1.0, n_total=None, eq4_weight=0.1, coef_eq=None):
    """alpha_e * sum_k c_k mean(w eq_k^2) and its parameter gradient.
    Also returns d(loss)/d(e) per point (seed of the entropy-net backward).
    n_total: global point count when this rank holds a shard (defaults to N)."""
    out, saved = forward4(params, x, y)
    eqs = residuals(out, Re, vis_t, e, scale)
    N = out.shape[0]
    nt = N if n_total is None else n_total
    ww = np.ones(N) if w is None else np.asarray(w).reshape(-1)
    c = [1.0, 1.0, 1.0, eq4_weight]
    sums = [float(np.sum(ww * q * q)) for q in eqs]
    if coef_eq is not None:      # explicit d loss / d(sum w eq_k^2 / 2) factors (what the C ABI takes)
        g = [coef_eq[k] * ww * eqs[k] for k in range(len(eqs))]
    else:
        g = [2.0 * alpha_e * c[k] * ww * eqs[k] / nt for k in range(len(eqs))]
    u, v = out[:, 0, 0], out[:, 1, 0]
    s, s2 = scale, scale * scale
    u_x, u_y = out[:, 0, 1] * s, out[:, 0, 2] * s
    v_x, v_y = out[:, 1, 1] * s, out[:, 1, 2] * s
    nu = 1.0 / Re + (0.0 if vis_t is None else np.asarray(vis_t).reshape(-1))
    g4 = g[3] if len(eqs) == 4 else np.zeros(N)
    r1 = g[0] + g4 * (u - 0.5)
    r2 = g[1] + g4 * (v - 0.5)
    r3 = g[2]
    adj = np.zeros_like(out)
    adj[:, 0, 0] = r1 * u_x + r2 * v_x + (g4 * eqs[0] if len(eqs) == 4 else 0.0)
    adj[:, 1, 0] = r1 * u_y + r2 * v_y + (g4 * eqs[1] if len(eqs) == 4 else 0.0)
    adj[:, 0, 1] = (r1 * u + r3) * s
    adj[:, 0, 2] = (r1 * v) * s
    adj[:, 1, 1] = (r2 * u) * s
    adj[:, 1, 2] = (r2 * v + r3) * s
    adj[:, 2, 1] = r1 * s
    adj[:, 2, 2] = r2 * s
    adj[:, 0, 3] = -nu * r1 * s2
    adj[:, 1, 3] = -nu * r2 * s2
    grad, zbars = backward4(params, x, y, saved, adj)
    return dict(out=out, saved=saved, eqs=eqs, sums=sums, grad=grad, zbars=zbars,
                out_adj=adj, e_adj=-g4)


def bc_loss_and_grad(params, x_b, y_b, u_b, v_b, alpha_b=1.0, n_total=None):
    out, saved = forward1(params, x_b, y_b)
    N = out.shape[0]
    nt = N if n_total is None else n_total
    du = out[:, 0] - np.asarray(u_b).reshape(-1)
    dv = out[:, 1] - np.asarray(v_b).reshape(-1)
    adj = np.zeros_like(out)
    adj[:, 0] = 2.0 * alpha_b * du / nt
    adj[:, 1] = 2.0 * alpha_b * dv / nt
    grad = backward1(params, x_b, y_b, saved, adj)
    return dict(pred=out, sums=[float(np.sum(du * du)), float(np.sum(dv * dv))], grad=grad)


def adam_step(p, g, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam (amsgrad=False, weight_decay=0, maximize=False) single
    tensor update, as used by NSFnet/pinn_solver.py:76-79,253."""
    m = b1 * m + (1.0 - b1) * g
    v = b2 * v + (1.0 - b2) * g * g
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    denom = np.sqrt(v) / np.sqrt(bc2) + eps
    p = p - (lr / bc1) * m / denom
    return p, m, v
