import os
import sys

import pytest

os.environ.setdefault("LTK_ALLOW_STANDIN", "1")   # headless stand-ins of the reference's base classes (livetalking_amd/hostshim.py)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` via gpurun)")


@pytest.fixture(scope="session")
def engine():
    """One engine on cuda:0 with the seeded synthetic Wav2Lip weights loaded."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from livetalking_amd.engine import Engine
    import synth_inputs as synth
    eng = Engine(0)
    eng.load_wav2lip(synth.wav2lip_state_dict(1234), max_frames=16)
    yield eng
    eng.close()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
