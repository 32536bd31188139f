"""`pinn_solver` module of NSFnet/ (pinn_solver.py:26-389) on the HIP engine."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "..")))
from nsfnet_amd.pinn_solver import PysicsInformedNeuralNetwork  # noqa: E402,F401
