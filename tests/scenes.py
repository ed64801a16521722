"""Seeded test scenes: moved to workloads/scenes.py (bench.py and smoke() use them too); re-exported for the tests."""
from workloads.scenes import *  # noqa: F401,F403
from workloads.scenes import make_camera_scene, make_scene, rel_err, torch_batch  # noqa: F401
