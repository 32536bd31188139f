"""``cavity_data.DataLoader`` with the reference's interface (NSFnet/cavity_data.py:23-105 and
the ev superset ev-NSFnet/cavity_data.py:25-161): boundary set, collocation sampling (+ optional
sort, SDF weights, [0,1] -> [-1,1] coordinate map) and DNS-field loading.  Host-side numpy;
nothing here is on the per-step hot path."""
import numpy as np
import scipy.io
from scipy.spatial import cKDTree

from .tools import LHSample, sort_pts


class DataLoader:
    def __init__(self, path=None, N_f=20000, N_b=1000, sort_training_points=True, sdf_weighting=None,
                 coord_transform=False):
        self.N_b = N_b
        self.x_min, self.x_max, self.y_min, self.y_max = 0.0, 1.0, 0.0, 1.0
        self.N_f = N_f
        self.pts_bc = None
        self.sort_training_points = sort_training_points
        self.sdf_config = sdf_weighting
        self.sdf_enabled = bool(getattr(sdf_weighting, 'enabled', False)) if sdf_weighting is not None else False
        self.sdf_weights = None
        self._bc_tree = None
        self.coord_transform = coord_transform
        self.coord_scale = 2.0 if coord_transform else 1.0

    def loading_boundary_data(self):
        """4 x 513 wall points in the order bottom, top (lid), left, right; regularised lid
        u = 1 - cosh(10 (x - 0.5)) / cosh(5), v = 0 everywhere (cavity_data.py:38-76)."""
        n_side, sharpness = 513, 10
        along_x = np.linspace(self.x_min, self.x_max, num=n_side)
        along_y = np.linspace(self.y_min, self.y_max, num=n_side)
        lid = 1 - np.cosh(sharpness * (along_x - 0.5)) / np.cosh(sharpness * 0.5)
        zeros = np.zeros([n_side])
        walls = [  # (x, y, u) per wall
            (along_x, self.y_min * np.ones([n_side]), zeros),
            (along_x, self.y_max * np.ones([n_side]), lid),
            (self.x_min * np.ones([n_side]), along_y, zeros),
            (self.x_max * np.ones([n_side]), along_y, zeros),
        ]
        x_b, y_b, u_b = (np.concatenate([wall[k] for wall in walls], axis=0).reshape([-1, 1]) for k in range(3))
        v_b = np.zeros_like(x_b)
        pts = np.hstack((x_b, y_b))
        if self.coord_transform:
            pts = self._to_centered_coords(pts)
            x_b, y_b = pts[:, 0:1], pts[:, 1:2]
            self.x_min, self.x_max, self.y_min, self.y_max = -1.0, 1.0, -1.0, 1.0
        self.pts_bc = pts
        if self.sdf_enabled:
            self._bc_tree = cKDTree(self.pts_bc)
        for line in ('-' * 29, 'N_train_bcs: %d' % x_b.shape[0], 'N_train_equ: %d' % self.N_f, '-' * 29):
            print(line)
        return x_b, y_b, u_b, v_b

    def loading_training_data(self):
        """LHS collocation points (+ sort by wall distance, + SDF weights) (cavity_data.py:78-92;
        ev :101-120).  As in the reference, the bounds already are [-1,1] when the coordinate map
        is on and the samples are mapped once more (ev :105-107)."""
        xye = LHSample(2, [[self.x_min, self.x_max], [self.y_min, self.y_max]], self.N_f)
        if self.coord_transform:
            xye = self._to_centered_coords(xye)
        if self.pts_bc is None:
            raise RuntimeError("need to load boundary data first!")
        if self.sort_training_points:
            xye, _ = sort_pts(xye, self.pts_bc)
        if self.sdf_enabled:
            self._compute_sdf_weights(xye)
        else:
            self.sdf_weights = None
        return xye[:, 0:1], xye[:, 1:2]

    def _compute_sdf_weights(self, pts):
        """w = min_w + (1 - min_w) exp(-decay d), d = distance to the nearest wall point, normalised to
        mean 1 and kept as float32 (ev :118-130)."""
        if self._bc_tree is None:
            self._bc_tree = cKDTree(self.pts_bc)
        cfg = self.sdf_config
        floor = min(max(float(getattr(cfg, 'min_weight', 0.2)) if cfg else 0.2, 1e-6), 1.0)
        rate = max(float(getattr(cfg, 'decay', 5.0)) if cfg else 5.0, 0.0)
        wall_dist = self._bc_tree.query(pts)[0]
        w = floor + (1.0 - floor) * np.exp(-rate * wall_dist)
        mean = np.mean(w)
        self.sdf_weights = (w / mean if mean > 0 else w).astype(np.float32)

    def get_sdf_weights(self):
        return self.sdf_weights

    def _to_centered_coords(self, pts):
        return pts * 2.0 - 1.0

    def _to_centered_values(self, values):
        return values * 2.0 - 1.0

    def get_coord_scale(self):
        return self.coord_scale

    def loading_evaluate_data(self, filename, with_pressure=None):
        """DNS reference fields X_ref, Y_ref, U_ref, V_ref[, P_ref] as (N,1) columns
        (cavity_data.py:94-105 returns 4 arrays, the ev flavour :144-161 returns 5).
        with_pressure=None picks the ev form when the coordinate-map / SDF options exist on this
        loader call site, i.e. when the file has P_ref and the caller asked for 5 values."""
        data = scipy.io.loadmat(filename)
        x, y, u, v = data['X_ref'], data['Y_ref'], data['U_ref'], data['V_ref']
        if self.coord_transform:
            x, y = self._to_centered_values(x), self._to_centered_values(y)
        cols = [a.reshape(-1, 1) for a in (x, y, u, v)]
        if with_pressure or (with_pressure is None and self._ev_flavour):
            cols.append(data['P_ref'].reshape(-1, 1))
        return tuple(cols)

    _ev_flavour = False


class EvDataLoader(DataLoader):
    """ev-NSFnet flavour: loading_evaluate_data returns (x, y, u, v, p)."""
    _ev_flavour = True
