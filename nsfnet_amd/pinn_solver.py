"""Drop-in ``pinn_solver.PysicsInformedNeuralNetwork`` (sic) for the plain NSFnet flavour.

Same constructor keywords, methods and attributes as the reference class
(NSFnet/pinn_solver.py:26-389) so NSFnet/train.py and NSFnet/test.py run unchanged; the
per-step work (``fwd_computing_loss_2d`` + ``loss.backward()`` + ``opt.step()``,
pinn_solver.py:197-278) is executed by the HIP pipeline through nsfnet_amd.engine.
There is no torch-autograd or CPU fallback.
"""
import os

import numpy as np
import scipy.io
import torch

from . import engine as _eng
from .net import FCNet


class AdamHandle:
    """What the scripts touch of torch.optim.Adam: ``param_groups[0]['lr']``
    (NSFnet/pinn_solver.py:235, ev-NSFnet/pinn_solver.py:437,497)."""

    def __init__(self, lr, betas=(0.9, 0.999), eps=1e-8):
        self.param_groups = [dict(lr=lr, betas=betas, eps=eps, weight_decay=0)]

    @property
    def lr(self):
        return self.param_groups[0]["lr"]


def _col(a):
    """numpy (N,1)/(N,) array or tensor -> contiguous float32 numpy vector."""
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(-1))


def default_device():
    if not torch.cuda.is_available():
        raise RuntimeError("nsfnet_amd needs an MI355X (ROCm) device: the PINN hot path is a HIP pipeline "
                           "with no CPU fallback")
    return torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))


class PysicsInformedNeuralNetwork:
    # training_type:  'unsupervised' | 'half-supervised'  (kept for signature compatibility)
    def __init__(self,
                 opt=None,
                 Re=1000,
                 layers=4,
                 hidden_size=120,
                 N_f=40000,
                 stage=0,
                 learning_rate=0.001,
                 weight_decay=0.9,          # accepted and ignored, as in the reference (:76-79)
                 outlet_weight=1,
                 bc_weight=1,
                 eq_weight=1,
                 ic_weight=1,
                 num_ins=2,
                 num_outs=3,
                 supervised_data_weight=1,
                 training_type='unsupervised',
                 net_params=None,
                 checkpoint_freq=10000,
                 checkpoint_path='./checkpoint/',
                 device=None):
        self.device = torch.device(device) if device is not None else default_device()
        self.Re = Re
        self.vis_t0 = 5.0 / self.Re
        self.checkpoint_freq = checkpoint_freq
        self.checkpoint_path = checkpoint_path
        self.layers = layers
        self.hidden_size = hidden_size
        self.N_f = N_f
        self.training_type = training_type
        self.stage = stage
        self.alpha_b = bc_weight
        self.alpha_e = eq_weight
        self.alpha_i = ic_weight
        self.alpha_o = outlet_weight
        self.alpha_s = supervised_data_weight
        self.loss_i = self.loss_o = self.loss_b = self.loss_e = self.loss_s = 0.0
        if num_outs != 3:
            raise ValueError("num_outs must be 3 (u, v, p)")

        self.net = self.initialize_NN(num_ins=num_ins, num_outs=num_outs, num_layers=layers,
                                      hidden_size=hidden_size)
        if net_params:
            self.net.load_state_dict(torch.load(net_params, map_location="cpu", weights_only=True))
        self.engine = _eng.PinnEngine(self.device, layers, hidden_size, Re, alpha_b=bc_weight, alpha_e=eq_weight,
                                      flavour="nsfnet", net=self.net.dev_net)
        self.opt = AdamHandle(learning_rate) if not opt else opt
        self.x_f = self.y_f = self.x_b = self.y_b = self.u_b = self.v_b = None
        self._terms = None
        self.log_every, self.save_every = 1000, 10000

    # ---------------------------------------------------------------- data
    def set_boundary_data(self, X=None, time=False):
        """X = (x_b, y_b, u_b, v_b), numpy (N,1) float64 (NSFnet/train.py:47-48; solver :82-90)."""
        self.x_b, self.y_b, self.u_b, self.v_b = (torch.as_tensor(_col(a)).reshape(-1, 1).to(self.device)
                                                  for a in X[:4])
        self.engine.set_boundary(_col(X[0]), _col(X[1]), _col(X[2]), _col(X[3]))

    def set_eq_training_data(self, X=None, time=False):
        """X = (x_f, y_f) collocation points (solver :92-99)."""
        self.x_f = torch.as_tensor(_col(X[0])).reshape(-1, 1).to(self.device)
        self.y_f = torch.as_tensor(_col(X[1])).reshape(-1, 1).to(self.device)
        self.engine.set_collocation(_col(X[0]), _col(X[1]))

    def set_optimizers(self, opt):
        self.opt = opt

    def set_stage(self, stage):
        self.stage = stage

    def initialize_NN(self, num_ins=2, num_outs=4, num_layers=6, hidden_size=160):
        return FCNet(num_ins=num_ins, num_outs=num_outs, num_layers=num_layers, hidden_size=hidden_size,
                     activation=torch.nn.Tanh, device=self.device)

    def set_eq_training_func(self, train_data_func):
        self.train_data_func = train_data_func

    # ---------------------------------------------------------------- model evaluation
    def neural_net_u(self, x, y):
        """u, v, p as (N,1) device tensors (solver :124-130)."""
        u, v, p = self.engine.predict(_col(x), _col(y))
        return u.reshape(-1, 1), v.reshape(-1, 1), p.reshape(-1, 1)

    def neural_net_equations(self, x, y):
        """eq1, eq2, eq3 at arbitrary points (solver :132-163), forward only."""
        plan = _eng.ResidualPlan(self.engine.net, _col(x), _col(y), with_backward=False)
        plan.forward(self.Re, save=False)
        return tuple(plan.field(k).reshape(-1, 1).clone() for k in ("eq1", "eq2", "eq3"))

    def predict(self, net_params, X):
        x, y = X
        return self.neural_net_u(x, y)

    # ---------------------------------------------------------------- loss / step
    def _publish_terms(self):
        mode = getattr(self, "_loss_mode_published", "MSE")
        self.engine.loss_mode = mode
        try:
            t = self.engine.loss_terms()
        finally:
            self.engine.loss_mode = "MSE"
        self._loss_mode_published = "MSE"
        f = self.engine.plan_f
        self.loss_eq1, self.loss_eq2, self.loss_eq3 = t["loss_eq1"], t["loss_eq2"], t["loss_eq3"]
        self.loss_e, self.loss_b, self.loss = t["loss_e"], t["loss_b"], t["loss"]
        self.eq1_pred, self.eq2_pred, self.eq3_pred = (f.field(k).reshape(-1, 1) for k in ("eq1", "eq2", "eq3"))
        b = self.engine.plan_b
        self.u_pred_b, self.v_pred_b = b.pred[0].reshape(-1, 1), b.pred[1].reshape(-1, 1)
        return t

    def fwd_computing_loss_2d(self, loss_mode='MSE'):
        """Loss of the current parameters AND its parameter gradient (the HIP pipeline fuses
        what the reference splits into this call and ``loss.backward()``, solver :197-226,252).
        Returns (loss, [loss_e, loss_b]) as 0-dim device tensors."""
        if loss_mode not in ('MSE', 'L2'):
            raise ValueError("loss_mode must be 'MSE' or 'L2'")
        assert self.x_f is not None and self.y_f is not None
        # 'L2' (solver :202-204, :214-217; no script of the reference selects it): 2-norms of the residual and
        # boundary-misfit vectors instead of mean squares - same kernels, other adjoint coefficients (engine.loss_mode)
        self.engine.loss_mode = loss_mode
        try:
            self.engine.loss_and_grad()
        finally:
            self.engine.loss_mode = 'MSE'
            self._loss_mode_published = loss_mode
        self._publish_terms()
        return self.loss, [self.loss_e, self.loss_b]

    def train(self, num_epoch=1, lr=1e-4, optimizer=None, scheduler=None, batchsize=None):
        self.opt.param_groups[0]['lr'] = lr
        return self.solve_Adam(self.fwd_computing_loss_2d, num_epoch, batchsize, scheduler)

    def solve_Adam(self, loss_func, num_epoch=1000, batchsize=None, scheduler=None):
        """The reference loop (solver :240-278): loss -> backward -> Adam step; log every 1000,
        checkpoint every 10000 (incl. step 0).  Adam moments persist across calls."""
        fused = getattr(loss_func, "__func__", None) is PysicsInformedNeuralNetwork.fwd_computing_loss_2d
        epoch_id = 0
        print('--------')
        print(num_epoch)
        print('--------')
        while epoch_id < num_epoch:
            lr = self.opt.param_groups[0]['lr']
            log_now = self.log_every and epoch_id % self.log_every == 0
            save_now = self.save_every and epoch_id % self.save_every == 0
            if fused and not (log_now or save_now):
                self.engine.step(lr)                 # no host sync; one hipGraph replay on a single GPU
            else:
                loss, losses = loss_func()
                self.engine.adam_step(lr)
            if scheduler:
                scheduler.step()
            if log_now:
                self.print_log(self.loss, [self.loss_e, self.loss_b], epoch_id, num_epoch)
            if save_now:
                self.save('model_cavity_loop_%d.pth' % epoch_id, N_HLayer=self.layers, N_neu=self.hidden_size,
                          N_f=self.N_f)
            epoch_id += 1
        self._publish_terms()

    def print_log(self, loss, losses, epoch_id, num_epoch):
        print("current lr is: %.6f " % (self.opt.param_groups[0]['lr']),
              "epoch/num_epoch: ", epoch_id + 1, "/", num_epoch,
              "eq1_loss: %.3e " % (self.loss_eq1.item()),
              "eq2_loss: %.3e " % (self.loss_eq2.item()),
              "eq4_loss: %.3e \n" % (self.loss_eq3.item()))

    # ---------------------------------------------------------------- evaluation / io
    def _errors(self, x, y, u, v):
        u_pred, v_pred, p_pred = (t.cpu().numpy().reshape(-1, 1) for t in self.neural_net_u(x, y))
        u_test, v_test = np.asarray(u).reshape(-1, 1), np.asarray(v).reshape(-1, 1)
        error_u = np.linalg.norm(u_test - u_pred, 2) / np.linalg.norm(u_test, 2)
        error_v = np.linalg.norm(v_test - v_pred, 2) / np.linalg.norm(v_test, 2)
        return error_u, error_v, u_pred, v_pred, p_pred

    def evaluate(self, x, y, u, v):
        """Relative L2 errors on the DNS grid (solver :308-325)."""
        error_u, error_v, *_ = self._errors(x, y, u, v)
        print('------------------------')
        print('Error u: %e' % (error_u))
        print('Error v: %e' % (error_v))
        return error_u, error_v

    def test(self, x, y, u, v, loop=None):
        """Errors + savemat of the predicted fields (solver :327-357).  The grid shape is taken
        from the inputs (the reference hard-codes 257x257, which breaks on the 385^2 Re4000 file)."""
        error_u, error_v, u_pred, v_pred, p_pred = self._errors(x, y, u, v)
        print('------------------------')
        print('Error u: %e' % (error_u))
        print('Error v: %e' % (error_v))
        print('------------------------')
        shape = np.asarray(x).shape if np.asarray(x).ndim == 2 and np.asarray(x).shape[1] > 1 else None
        if shape is None:
            side = int(round(np.sqrt(u_pred.size)))
            shape = (side, side) if side * side == u_pred.size else (u_pred.size, 1)
        scipy.io.savemat('cavity_result_loop_%d.mat' % (loop),
                         {'U_pred': u_pred.reshape(shape), 'V_pred': v_pred.reshape(shape),
                          'P_pred': p_pred.reshape(shape), 'lam_bcs': self.alpha_b, 'lam_equ': self.alpha_e})
        return error_u, error_v

    def save(self, filename, directory=None, N_HLayer=None, N_neu=None, N_f=None, lr=None):
        """Reference checkpoint layout (solver :359-380):
        results/Re{Re}/{L}x{H}_Nf{N/1000}k_lamB{alpha_b}{stage}/<filename> = net.state_dict()."""
        Re_folder = 'Re' + str(self.Re)
        NNsize = str(N_HLayer) + 'x' + str(N_neu) + '_Nf' + str(np.int32(N_f / 1000)) + 'k'
        lambdas = 'lamB' + str(self.alpha_b)
        relative_path = '/results/' + Re_folder + '/' + NNsize + '_' + lambdas + str(self.stage) + '/'
        if not directory:
            directory = os.getcwd()
        save_results_to = directory + relative_path
        os.makedirs(save_results_to, exist_ok=True)
        torch.save(self.net.state_dict(), save_results_to + filename)
        save_matlab_to = directory + '/loss/'
        os.makedirs(save_matlab_to, exist_ok=True)
        if getattr(self, "loss_eq1", None) is not None:
            scipy.io.savemat(save_matlab_to + 'eq_losses.mat',
                             {'eq1': self.loss_eq1.item(), 'eq2': self.loss_eq2.item(), 'eq3': self.loss_eq3.item()})

    def divergence(self, x_star, y_star):
        return self.neural_net_equations(x_star, y_star)[2]
