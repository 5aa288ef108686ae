"""pointfoot_amd — MI355X-native drop-in for the legged_gym `pointfoot` hot path.

`LEGGED_GYM_ROOT_DIR` keeps the reference's meaning (legged_gym/__init__.py:33-34): the directory that
holds `resources/` and `logs/`.  Here that is the package directory itself.
"""
import os

LEGGED_GYM_ROOT_DIR = os.path.dirname(os.path.realpath(__file__))
LEGGED_GYM_ENVS_DIR = os.path.join(LEGGED_GYM_ROOT_DIR, 'envs')
