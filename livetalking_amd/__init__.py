"""MI355X-native lip-sync render hot path behind LiveTalking's avatar plugin surface.

`livetalking_amd.engine.Engine` wraps libltk_hip.so (hand-written HIP for gfx950);
`livetalking_amd.avatars.wav2lip_avatar` mirrors the reference plugin module
(avatars/wav2lip_avatar.py) and routes every per-frame op to the engine.
"""
__all__ = ["engine"]
