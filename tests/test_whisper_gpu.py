"""HIP Whisper audio-feature path (ltk_whisper_step) against the installed transformers WhisperFeatureExtractor +
WhisperModel encoder (the library the reference calls, audio2feature.py:20-23) and the reference's chunk slicing.

Tolerances: log-mel input features max-abs <= 2e-3 (fp16 storage of values in [-1.5, 1.5]); hidden states relative
L2 <= 1e-2 (fp16 activations vs fp32); chunks likewise."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytest.importorskip("transformers")

import synth_inputs as synth  # noqa: E402
from oracle import whisper_oracle as WO  # noqa: E402


@pytest.mark.gpu
def test_whisper_features_vs_transformers():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from livetalking_amd.engine import Engine
    model = WO.tiny_whisper(0)
    eng = Engine(0)
    eng.load_whisper({k: v for k, v in model.encoder.state_dict().items()})
    B, l = 16, 10
    wav = synth.synthetic_audio(2.0)[: (20 + 2 * B) * 320]          # the 52-chunk buffer of one step
    feat_arr, hs, feats = WO.audio2feat(model, wav)
    d_out = torch.zeros(B, 50, 384, dtype=torch.float32, device="cuda")
    eng.whisper_step(wav, B, first_row=int((0 + l / 2) * 2), d_out_ptr=d_out.data_ptr())
    got_in = eng.whisper_debug_get("input_features", (80, 3000))
    ref_in = feats[0].numpy()
    e_in = float(np.abs(got_in - ref_in).max())
    print(f"[whisper] input_features max abs err {e_in:.3e} (range {ref_in.min():.2f}..{ref_in.max():.2f})")
    assert e_in <= 2e-3
    for i in range(5):
        got = eng.whisper_debug_get(f"hidden_states.{i}", (384, 1500))
        ref = hs[i].numpy().T
        r = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
        print(f"[whisper] hidden_states.{i} rel_l2={r:.3e} refmax={np.abs(ref).max():.3g}")
        assert r <= 1e-2
    ref_chunks = np.stack(WO.feature2chunks(feat_arr, B, l))
    got_chunks = d_out.cpu().numpy()
    r = float(np.linalg.norm(got_chunks - ref_chunks) / np.linalg.norm(ref_chunks))
    print(f"[whisper] chunks rel_l2={r:.3e}")
    assert got_chunks.shape == ref_chunks.shape == (B, 50, 384) and r <= 1e-2
    eng.close()
