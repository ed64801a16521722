"""Hybrid-tracer scenes: moved to workloads/playground_scenes.py; re-exported for the tests."""
from workloads.playground_scenes import *  # noqa: F401,F403
