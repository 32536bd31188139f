"""`cavity_data` module of NSFnet/ (cavity_data.py:23-105)."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "..")))
from nsfnet_amd.cavity_data import DataLoader  # noqa: E402,F401
