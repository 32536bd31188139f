"""Drop-in ``pinn_solver.PysicsInformedNeuralNetwork`` for the ev-NSFnet flavour
(entropy-viscosity regularised PINN, one process per GPU).

Same constructor keywords, methods and attributes as the reference class
(ev-NSFnet/pinn_solver.py:27-765); ev-NSFnet/train.py and test.py drive it unchanged.
What differs is where the arithmetic happens:
  * loss + gradient + Adam run in the HIP pipeline (nsfnet_amd.engine), no autograd;
  * the lagged artificial viscosity state vis_t_minus stays ON THE DEVICE (the reference
    round-trips it through host numpy every step, :327-334);
  * data parallelism = ONE RCCL all-reduce of [grads | loss sums] per step instead of two
    DDP reducers plus 2-3 blocking scalar all-reduces (:105-106, :414-424).  Gradients and
    losses are normalised by the GLOBAL point counts (mathematically clean full-batch
    gradient; the reference's extra 1/W factor on local gradients only acts through
    Adam's eps - see DESIGN.md).
"""
import os
import time

import numpy as np
import scipy.io
import torch
import torch.distributed as dist

from . import engine as _eng
from .net import FCNet
from .pinn_solver import AdamHandle, _col, default_device


class PysicsInformedNeuralNetwork:
    tb_writer = None
    global_step = 0

    def __init__(self,
                 opt=None,
                 Re=1000,
                 layers=6,
                 layers_1=6,
                 hidden_size=80,
                 hidden_size_1=20,
                 N_f=100000,
                 alpha_evm=0.03,
                 learning_rate=0.001,
                 weight_decay=0.9,
                 outlet_weight=1,
                 bc_weight=10,
                 eq_weight=1,
                 ic_weight=0.1,
                 num_ins=2,
                 num_outs=3,
                 num_outs_1=1,
                 supervised_data_weight=1,
                 net_params=None,
                 net_params_1=None,
                 checkpoint_freq=2000,
                 checkpoint_path='./checkpoint/'):
        # torchrun env contract (ev-NSFnet/pinn_solver.py:57-63)
        self.rank = int(os.environ.get('RANK', 0))
        self.local_rank = int(os.environ.get('LOCAL_RANK', 0))
        self.world_size = int(os.environ.get('WORLD_SIZE', 1))
        self.device = default_device()
        torch.cuda.set_device(self.device)

        self.evm = None
        self.Re = Re
        self.vis_t0 = 20.0 / self.Re
        self.layers, self.layers_1 = layers, layers_1
        self.hidden_size, self.hidden_size_1 = hidden_size, hidden_size_1
        self.N_f = N_f
        self.current_stage = ' '
        self.checkpoint_freq, self.checkpoint_path = checkpoint_freq, checkpoint_path
        self.alpha_evm = alpha_evm
        self.alpha_b, self.alpha_e = bc_weight, eq_weight
        self.alpha_i, self.alpha_o = ic_weight, outlet_weight
        self.alpha_s = supervised_data_weight
        self.loss_i = self.loss_o = self.loss_b = self.loss_e = self.loss_s = 0.0
        self.x_s = self.y_s = self.u_s = self.v_s = self.p_s = None
        self._p_mask = None
        self.supervision_point_count = 0
        self.supervision_total_points = 0
        self.supervision_has_data = False
        self.supervision_enabled = False
        self.eq_weights = None
        self.coord_scale = 1.0
        self.coord_scale_sq = 1.0
        if num_outs != 3 or num_outs_1 != 1:
            raise ValueError("num_outs must be 3 (u,v,p) and num_outs_1 must be 1 (e)")

        self.net = self.initialize_NN(num_ins=num_ins, num_outs=num_outs, num_layers=layers, hidden_size=hidden_size)
        self.net_1 = self.initialize_NN(num_ins=num_ins, num_outs=num_outs_1, num_layers=layers_1,
                                        hidden_size=hidden_size_1)
        self.is_distributed = dist.is_available() and dist.is_initialized() and self.world_size > 1
        if self.is_distributed:
            # DDP's wrap-time broadcast from rank 0 (:105-106): make the replicas identical
            for n in (self.net, self.net_1):
                dist.broadcast(n.dev_net.params, src=0)
                n.dev_net.prepare()
        if net_params:
            if self.rank == 0:
                print(f"Loading net params from {net_params}")
            self.net.load_state_dict(torch.load(net_params, map_location="cpu", weights_only=True))
        if net_params_1:
            if self.rank == 0:
                print(f"Loading net_1 params from {net_params_1}")
            self.net_1.load_state_dict(torch.load(net_params_1, map_location="cpu", weights_only=True))

        self.engine = _eng.PinnEngine(
            self.device, layers, hidden_size, Re, alpha_b=bc_weight, alpha_e=eq_weight, flavour="ev",
            n_hidden_e=layers_1, hidden_e=hidden_size_1, alpha_evm=alpha_evm, alpha_s=0.0,
            process_group=(dist.group.WORLD if self.is_distributed else None),
            world_size=(self.world_size if self.is_distributed else 1),
            net=self.net.dev_net, net_e=self.net_1.dev_net)
        self.opt = AdamHandle(learning_rate) if not opt else opt
        self.x_f = self.y_f = self.x_b = None
        self._stash_bc = None

        if self.rank == 0:
            print(f"Distributed training setup:")
            print(f"  World size: {self.world_size}")
            print(f"  Rank: {self.rank}")
            print(f"  Local rank: {self.local_rank}")
            print(f"  Device: {self.device}")

    # ---------------------------------------------------------------- data
    def _shard(self, total):
        """Contiguous blocks, last rank takes the remainder (:144-147, :165-168)."""
        w = self.world_size if self.is_distributed else 1
        r = self.rank if self.is_distributed else 0
        if total < w:
            # the reference would hand ranks 0..w-2 an empty block and divide by zero in their means; the
            # count is global, so every rank raises here together instead of one rank leaving a collective
            raise ValueError("%d points cannot be sharded over %d ranks" % (total, w))
        per = total // w
        lo = r * per
        hi = lo + per if r < w - 1 else total
        return lo, hi

    def set_boundary_data(self, X=None, time=False):
        total = X[0].shape[0]
        lo, hi = self._shard(total)
        xb, yb, ub, vb = (_col(a)[lo:hi] for a in X[:4])
        self.x_b, self.y_b, self.u_b, self.v_b = (torch.as_tensor(a).reshape(-1, 1).to(self.device)
                                                  for a in (xb, yb, ub, vb))
        self.engine.set_boundary(xb, yb, ub, vb, n_global=total)
        if self.rank == 0:
            print(f"GPU {self.rank}: Processing {hi - lo} boundary points out of {total} total")

    def set_eq_training_data(self, X=None, time=False, weights=None):
        total = X[0].shape[0]
        lo, hi = self._shard(total)
        xf, yf = _col(X[0])[lo:hi], _col(X[1])[lo:hi]
        self.x_f = torch.as_tensor(xf).reshape(-1, 1).to(self.device)
        self.y_f = torch.as_tensor(yf).reshape(-1, 1).to(self.device)
        w = None
        if weights is not None:
            w = _col(weights)[lo:hi]
            self.eq_weights = torch.as_tensor(w).to(self.device)
        else:
            self.eq_weights = None
        self.engine.alpha_evm = float(self.alpha_evm)
        self.engine.set_collocation(xf, yf, weights=w, n_global=total)     # includes init_vis_t (:184)
        if self.rank == 0:
            print(f"GPU {self.rank}: Processing {hi - lo} equation points out of {total} total")

    def init_vis_t(self):
        self.engine.alpha_evm = float(self.alpha_evm)
        self.engine.init_vis_t()

    def set_coordinate_transform(self, scale: float):
        if scale is None or scale <= 0:
            self.coord_scale = 1.0
        else:
            self.coord_scale = float(scale)
        self.coord_scale_sq = self.coord_scale ** 2
        self.engine.scale = self.coord_scale

    def clear_supervised_data(self):
        self.x_s = self.y_s = self.u_s = self.v_s = self.p_s = None
        self._p_mask = None
        self.supervision_point_count = 0
        self.supervision_total_points = 0
        self.supervision_has_data = False
        self.supervision_enabled = False
        self.engine.set_supervised(None, None, None, None)

    def set_supervised_data(self, data):
        """(x, y, u, v, p) supervised samples; split with np.array_split over ranks (:219-221);
        NaN pressure targets are masked (:247-251, :405-410)."""
        if data is None:
            self.clear_supervised_data()
            return
        x, y, u, v, p = data
        x, y, u, v = (np.asarray(a) for a in (x, y, u, v))
        p = np.asarray(p) if p is not None else None
        total = x.shape[0]
        self.supervision_total_points = int(total)
        if total == 0:
            self.clear_supervised_data()
            return
        if self.is_distributed:
            idx = np.array_split(np.arange(total), self.world_size)[self.rank]
        else:
            idx = np.arange(total)
        xs, ys, us, vs = (_col(a)[idx] for a in (x, y, u, v))
        ps = _col(p)[idx] if p is not None else None
        self.supervision_point_count = int(xs.shape[0])
        to_t = lambda a: None if a is None else torch.as_tensor(a).reshape(-1, 1).to(self.device)
        self.x_s, self.y_s, self.u_s, self.v_s, self.p_s = (to_t(a) for a in (xs, ys, us, vs, ps))
        self._p_mask = None if ps is None else torch.isfinite(self.p_s)
        self.engine.set_supervised(xs, ys, us, vs, ps, n_global=total)    # an empty local share is allowed (:400)
        self.supervision_has_data = self.supervision_total_points > 0
        self.supervision_enabled = self.supervision_has_data and self.alpha_s != 0.0
        self.engine.alpha_s = float(self.alpha_s) if self.supervision_enabled else 0.0

    def set_supervised_loss_weight(self, weight: float):
        self.alpha_s = float(weight)
        self.supervision_enabled = self.supervision_has_data and self.alpha_s != 0.0
        self.engine.alpha_s = float(self.alpha_s) if self.supervision_enabled else 0.0

    def set_optimizers(self, opt):
        self.opt = opt

    def set_alpha_evm(self, alpha):
        self.alpha_evm = alpha
        self.engine.alpha_evm = float(alpha)

    def initialize_NN(self, num_ins=3, num_outs=3, num_layers=10, hidden_size=50):
        return FCNet(num_ins=num_ins, num_outs=num_outs, num_layers=num_layers, hidden_size=hidden_size,
                     activation=torch.nn.Tanh, device=self.device)

    def set_eq_training_func(self, train_data_func):
        self.train_data_func = train_data_func

    # ---------------------------------------------------------------- model evaluation
    def neural_net_u(self, x, y):
        """u, v as (N,), p, e as (N,1) device tensors - the reference's mixed shapes (:284-287)."""
        u, v, p, e = self.engine.predict(_col(x), _col(y), with_e=True)
        return u, v, p.reshape(-1, 1), e.reshape(-1, 1)

    def predict(self, net_params, X):
        x, y = X
        return self.neural_net_u(x, y)

    @property
    def vis_t(self):
        f = self.engine.plan_f
        return None if f is None else f.vis_t.reshape(-1, 1)

    @property
    def vis_t_minus(self):
        f = self.engine.plan_f
        return None if (f is None or f.vis_t_minus is None) else f.vis_t_minus.reshape(-1, 1)

    # ---------------------------------------------------------------- loss / step
    def _publish_terms(self):
        t = self.engine.loss_terms()
        f = self.engine.plan_f
        self.loss_eq1, self.loss_eq2, self.loss_eq3, self.loss_eq4 = (t["loss_eq%d" % k] for k in (1, 2, 3, 4))
        self.loss_e, self.loss_b, self.loss_s, self.loss = t["loss_e"], t["loss_b"], t["loss_s"], t["loss"]
        self.eq1_pred, self.eq2_pred, self.eq3_pred, self.eq4_pred = (
            f.field(k).reshape(-1, 1) for k in ("eq1", "eq2", "eq3", "eq4"))
        self.evm = self.engine.plan_e.pred[0].reshape(-1, 1)
        return t

    def fwd_computing_loss_2d(self, loss_mode='MSE'):
        """Global loss AND its gradient (fused; reference :372-428 + loss.backward() :469)."""
        if loss_mode != 'MSE':
            raise NotImplementedError("only the MSE loss is implemented")
        assert self.x_f is not None and self.y_f is not None
        self.engine.loss_and_grad()
        self._publish_terms()
        return self.loss, [self.loss_e, self.loss_b]

    def train(self, num_epoch=1, lr=1e-4, optimizer=None, scheduler=None, batchsize=None):
        if self.opt is not None:
            self.opt.param_groups[0]['lr'] = lr
        return self.solve_Adam(self.fwd_computing_loss_2d, num_epoch, batchsize, scheduler)

    def solve_Adam(self, loss_func, num_epoch=1000, batchsize=None, scheduler=None):
        """Reference loop :440-487 incl. its schedule: the entropy net trains for exactly one
        step in every 10 000 and Adam is re-created (moments reset) at every stage start and at
        steps k*10000 and k*10000+1."""
        self._last_log_time, self._last_log_epoch = time.time(), 0
        if not hasattr(self, 'log_interval'):
            self.log_interval = 100
        self.freeze_evm_net(0)
        fused = getattr(loss_func, "__func__", None) is PysicsInformedNeuralNetwork.fwd_computing_loss_2d
        for epoch_id in range(num_epoch):
            self.global_step += 1
            self._apply_freeze_schedule(epoch_id)
            interval = self.log_interval if self.log_interval > 0 else 100
            log_now = self.rank == 0 and (epoch_id == 0 or (epoch_id + 1) % interval == 0 or epoch_id == num_epoch - 1)
            save_now = self.rank == 0 and (epoch_id == 0 or epoch_id % 10000 == 0)
            if fused and not (log_now or save_now):
                self.engine.step(self.opt.param_groups[0]['lr'])
            else:
                loss, losses = loss_func()
                self.engine.adam_step(self.opt.param_groups[0]['lr'])
            if scheduler:
                scheduler.step()
            if log_now:
                self.print_log(self.loss, [self.loss_e, self.loss_b], epoch_id, num_epoch)
            if save_now:
                self.save('model_cavity_loop%d.pth' % (epoch_id), N_HLayer=self.layers, N_neu=self.hidden_size,
                          N_f=self.N_f)
        if num_epoch > 0:
            self._publish_terms()

    def _apply_freeze_schedule(self, epoch_id):
        """:459-462 - unfreeze at k*10000 (k >= 1), re-freeze one step later."""
        if epoch_id != 0 and epoch_id % 10000 == 0:
            self.defreeze_evm_net(epoch_id)
        if (epoch_id - 1) % 10000 == 0:
            self.freeze_evm_net(epoch_id)

    def freeze_evm_net(self, epoch_id):
        """:489-499 - entropy net frozen, fresh Adam over the main net."""
        self.engine.e_trainable = False
        self.engine.net.reset_adam()
        self.opt = AdamHandle(self.opt.param_groups[0]['lr'])

    def defreeze_evm_net(self, epoch_id):
        """:501-511 - entropy net trainable, fresh Adam over both nets."""
        self.engine.e_trainable = True
        self.engine.net.reset_adam()
        self.engine.net_e.reset_adam()
        self.opt = AdamHandle(self.opt.param_groups[0]['lr'])

    # ---------------------------------------------------------------- logging
    def print_log(self, loss, losses, epoch_id, num_epoch):
        """Rank-0 progress line.  Kept from the reference's block (:513-650) only what defines a number
        someone compares: throughput = steps/s since the last log x local (collocation + boundary) points
        (:581-591) and Re_eff = 1 / (1/Re + mean vis_t) (:566-568)."""
        now = time.time()
        steps = max(epoch_id - getattr(self, '_last_log_epoch', 0), 1)
        dt = now - getattr(self, '_last_log_time', now)
        rate = steps / dt if dt > 0 else 0.0
        pts = sum(t.shape[0] for t in (self.x_f, self.x_b) if t is not None)
        vis = self.vis_t
        Re_eff = 1.0 / (1.0 / self.Re + vis.mean().item()) if vis is not None else float('nan')
        terms = ' '.join('%s=%.3e' % (k, float(getattr(self, 'loss_' + k))) for k in ('eq1', 'eq2', 'eq3', 'eq4'))
        print('[%s] %d/%d  loss=%.4e  eq_total=%.3e boundary=%.3e  %s' % (
            self.current_stage, epoch_id + 1, num_epoch, float(self.loss), float(self.loss_e), float(self.loss_b), terms))
        if self.supervision_total_points > 0 and self.alpha_s != 0.0:
            print('  supervision: loss=%.3e alpha=%.3g samples_total=%d local=%d' % (
                float(self.loss_s), self.alpha_s, self.supervision_total_points, self.supervision_point_count))
        print('  it/s=%.2f  throughput=%.1f pts/s  lr=%.2e  Re=%s  Re_eff=%.1f  alpha_evm=%s' % (
            rate, rate * pts, self.opt.param_groups[0]['lr'], self.Re, Re_eff, self.alpha_evm))
        self._last_log_time, self._last_log_epoch = now, epoch_id

    # ---------------------------------------------------------------- evaluation / io
    def _errors(self, x, y, u, v, p):
        u_pred, v_pred, p_pred, e_pred = (t.cpu().numpy().reshape(-1, 1) for t in self.neural_net_u(x, y))
        u_test, v_test, p_test = (np.asarray(a).reshape(-1, 1) for a in (u, v, p))
        mask_p = ~np.isnan(p_test)
        error_u = 100 * np.linalg.norm(u_test - u_pred, 2) / np.linalg.norm(u_test, 2)
        error_v = 100 * np.linalg.norm(v_test - v_pred, 2) / np.linalg.norm(v_test, 2)
        error_p = 100 * np.linalg.norm(p_test[mask_p] - p_pred[mask_p], 2) / np.linalg.norm(p_test[mask_p], 2)
        return error_u, error_v, error_p, u_pred, v_pred, p_pred, e_pred

    def evaluate(self, x, y, u, v, p):
        """Relative L2 errors in percent (:669-693)."""
        error_u, error_v, error_p, *_ = self._errors(x, y, u, v, p)
        if self.rank == 0:
            print('------------------------')
            print('Error u: %.2f %%' % (error_u))
            print('Error v: %.2f %%' % (error_v))
            print('Error p: %.2f %%' % (error_p))
        return error_u, error_v, error_p

    def test(self, x, y, u, v, p, loop=None, save_dir='./results/Re5000/test_result'):
        """Errors + savemat (:695-740); grid shape taken from the inputs instead of 257x257."""
        error_u, error_v, error_p, u_pred, v_pred, p_pred, e_pred = self._errors(x, y, u, v, p)
        if self.rank == 0:
            print('------------------------')
            print('Error u: %.3f %%' % (error_u))
            print('Error v: %.3f %%' % (error_v))
            print('Error p: %.3f %%' % (error_p))
            print('------------------------')
            xa = np.asarray(x)
            if xa.ndim == 2 and xa.shape[1] > 1:
                shape = xa.shape
            else:
                side = int(round(np.sqrt(u_pred.size)))
                shape = (side, side) if side * side == u_pred.size else (u_pred.size, 1)
            os.makedirs(save_dir, exist_ok=True)
            scipy.io.savemat(os.path.join(save_dir, f'cavity_result_loop_{loop}.mat'),
                             {'U_pred': u_pred.reshape(shape), 'V_pred': v_pred.reshape(shape),
                              'P_pred': p_pred.reshape(shape), 'E_pred': e_pred.reshape(shape),
                              'error_u': error_u, 'error_v': error_v, 'error_p': error_p,
                              'lam_bcs': self.alpha_b, 'lam_equ': self.alpha_e})
        return error_u, error_v, error_p

    def save(self, filename, directory=None, N_HLayer=None, N_neu=None, N_f=None):
        """Reference layout (:742-759): results/Re{Re}/{L}x{H}_Nf{N/1000}k_lamB{ab}_alpha{a}{stage}/
        <filename> (main net state_dict) and <filename>_evm (entropy net)."""
        Re_folder = 'Re' + str(self.Re)
        NNsize = str(N_HLayer) + 'x' + str(N_neu) + '_Nf' + str(np.int32(N_f / 1000)) + 'k'
        lambdas = 'lamB' + str(self.alpha_b) + '_alpha' + str(self.alpha_evm) + str(self.current_stage)
        relative_path = '/results/' + Re_folder + '/' + NNsize + '_' + lambdas + '/'
        if not directory:
            directory = os.getcwd()
        save_results_to = directory + relative_path
        os.makedirs(save_results_to, exist_ok=True)
        torch.save(self.net.state_dict(), save_results_to + filename)
        torch.save(self.net_1.state_dict(), save_results_to + filename + '_evm')

    def neural_net_equations(self, x, y):
        """eq1..eq4 at arbitrary points, forward only, with vis_t = vis_t0 as the reference does
        when no lagged state exists (:330-331)."""
        xs, ys = _col(x), _col(y)
        pe = _eng.ValuePlan(self.engine.net_e, xs, ys, with_backward=False)
        pe.forward(save=False)
        plan = _eng.ResidualPlan(self.engine.net, xs, ys, with_backward=False)
        plan.vis_t_minus = torch.full((plan.n,), float("inf"), dtype=torch.float32, device=self.device)
        plan.forward(self.Re, e=pe.pred[0], vis_t0=self.vis_t0, alpha_evm=self.alpha_evm, scale=self.coord_scale,
                     save=False)
        return tuple(plan.field(k).reshape(-1, 1).clone() for k in ("eq1", "eq2", "eq3", "eq4"))

    def divergence(self, x_star, y_star):
        return self.neural_net_equations(x_star, y_star)[2]
