"""YAML-backed run configuration with the attribute tree ev-NSFnet/train.py reads
(config.py:9-178: cfg.physics / cfg.network / cfg.training{.sdf_weighting,.training_stages} /
cfg.supervision, ConfigManager.from_file / print_config / validate_config)."""
from dataclasses import dataclass, field, fields, is_dataclass
from typing import List

import yaml


@dataclass
class PhysicsConfig:
    Re: int = 5000
    alpha_evm: float = 0.05
    bc_weight: float = 10.0
    eq_weight: float = 1.0


@dataclass
class NetworkConfig:
    layers: int = 6
    layers_1: int = 4
    hidden_size: int = 80
    hidden_size_1: int = 40


@dataclass
class TrainingStage:
    alpha: float
    epochs: int
    lr: float
    name: str


@dataclass
class SupervisionConfig:
    enabled: bool = False
    num_samples: int = 0
    loss_weight: float = 1.0


@dataclass
class SDFWeightConfig:
    enabled: bool = False
    min_weight: float = 0.2
    decay: float = 5.0


def _default_stages():
    table = [(0.05, 1e-3), (0.03, 2e-4), (0.01, 4e-5), (0.005, 1e-5), (0.002, 2e-6), (0.002, 2e-6)]
    return [TrainingStage(a, 500000, lr, "Stage %d" % (i + 1)) for i, (a, lr) in enumerate(table)]


@dataclass
class TrainingConfig:
    N_f: int = 120000
    log_interval: int = 1000
    enable_tensorboard: bool = True
    tb_log_dir: str = "runs"
    sort_training_points: bool = True
    sdf_weighting: SDFWeightConfig = field(default_factory=SDFWeightConfig)
    coordinate_transform: bool = False
    training_stages: List[TrainingStage] = field(default_factory=_default_stages)


@dataclass
class AppConfig:
    experiment_name: str = "NSFnet_MI355X"
    description: str = ""
    physics: PhysicsConfig = field(default_factory=PhysicsConfig)
    network: NetworkConfig = field(default_factory=NetworkConfig)
    training: TrainingConfig = field(default_factory=TrainingConfig)
    supervision: SupervisionConfig = field(default_factory=SupervisionConfig)


def _fill(obj, data):
    """Copy known keys of a (nested) dict onto a dataclass instance; unknown keys are ignored."""
    if not isinstance(data, dict):
        return obj
    known = {f.name: f for f in fields(obj)}
    for key, val in data.items():
        if key not in known:
            continue
        cur = getattr(obj, key)
        if key == "training_stages":
            setattr(obj, key, [TrainingStage(float(s["alpha"]), int(s["epochs"]), float(s["lr"]), str(s["name"]))
                               for s in (val or [])])
        elif is_dataclass(cur):
            _fill(cur, val)
        else:
            setattr(obj, key, type(cur)(val) if cur is not None and not isinstance(cur, str) else val)
    return obj


class ConfigManager:
    def __init__(self, config=None):
        self.config = config or AppConfig()

    @classmethod
    def from_file(cls, path):
        with open(path, "r", encoding="utf-8") as fh:
            raw = yaml.safe_load(fh) or {}
        mgr = cls(_fill(AppConfig(), raw))
        mgr.validate_config()
        return mgr

    def validate_config(self):
        c = self.config
        problems = []
        if c.physics.Re <= 0:
            problems.append("physics.Re must be positive")
        if min(c.network.layers, c.network.layers_1, c.network.hidden_size, c.network.hidden_size_1) < 1:
            problems.append("network sizes must be >= 1")
        if c.training.N_f < 1:
            problems.append("training.N_f must be >= 1")
        for st in c.training.training_stages:
            if st.epochs < 0 or st.lr <= 0:
                problems.append("stage %s: epochs >= 0 and lr > 0 required" % st.name)
        if problems:
            raise ValueError("invalid configuration: " + "; ".join(problems))
        return True

    def print_config(self):
        c = self.config
        print("experiment : %s" % c.experiment_name)
        print("physics    : Re=%s alpha_evm=%s bc_weight=%s eq_weight=%s"
              % (c.physics.Re, c.physics.alpha_evm, c.physics.bc_weight, c.physics.eq_weight))
        print("network    : %dx%d (u,v,p) + %dx%d (e)"
              % (c.network.layers, c.network.hidden_size, c.network.layers_1, c.network.hidden_size_1))
        t = c.training
        print("training   : N_f=%d log_interval=%d sort=%s sdf=%s coord_transform=%s stages=%d"
              % (t.N_f, t.log_interval, t.sort_training_points, t.sdf_weighting.enabled, t.coordinate_transform,
                 len(t.training_stages)))
        print("supervision: enabled=%s samples=%d weight=%s"
              % (c.supervision.enabled, c.supervision.num_samples, c.supervision.loss_weight))
