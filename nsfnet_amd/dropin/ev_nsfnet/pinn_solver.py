"""`pinn_solver` module of ev-NSFnet/ (pinn_solver.py:27-765) on the HIP engine."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "..")))
from nsfnet_amd.ev_pinn_solver import PysicsInformedNeuralNetwork  # noqa: E402,F401
