"""Seeded synthetic inputs: re-export of livetalking_amd.synth (the generators hold no
reference arithmetic; they live product-side so bench.py's timed path imports nothing from
oracle/).  TEST INFRASTRUCTURE (see oracle/__init__.py)."""
from livetalking_amd.synth import *  # noqa: F401,F403
from livetalking_amd.synth import OUTPUT_HEAD_PREFIX, synthetic_audio, wav2lip_avatar, wav2lip_state_dict  # noqa: F401
