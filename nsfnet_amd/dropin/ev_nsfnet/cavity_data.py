"""`cavity_data` module of ev-NSFnet/ (cavity_data.py:25-161): evaluate data comes with P_ref."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "..")))
from nsfnet_amd.cavity_data import EvDataLoader as DataLoader  # noqa: E402,F401
