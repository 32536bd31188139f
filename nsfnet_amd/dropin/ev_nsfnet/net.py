"""`net` module of the reference (net.py:22-54), backed by the HIP engine."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "..")))
from nsfnet_amd.net import FCNet  # noqa: E402,F401
