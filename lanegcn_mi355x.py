"""Top-level plugin module so the reference's drivers can load this build by name:
    python train.py -m lanegcn_mi355x      (train.py:63-64: import_module(args.model).get_model())"""
import lanegcn_amd  # noqa: F401  (import shim for the lanegcn-1_amd/ package directory)
from lanegcn_amd.lanegcn import *  # noqa: F401,F403
from lanegcn_amd.lanegcn import config, get_model  # noqa: F401
