"""The seeded synthetic workload generators live in tools/synth.py (bench.py and smoke() use them too: VERDICT r5 hygiene #9); the tests keep this name."""
from tools.synth import *  # noqa: F401,F403
