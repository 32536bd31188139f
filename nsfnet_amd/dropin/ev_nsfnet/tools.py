"""`tools` module of the reference (tools.py:30-83), vectorised."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "..")))
from nsfnet_amd.tools import LHSample, distance, minDistance, sort_pts, min_distances  # noqa: E402,F401
