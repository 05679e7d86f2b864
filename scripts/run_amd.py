#!/usr/bin/env python3
"""Launcher that drops this engine behind an unmodified LiveTalking checkout.

    cd /path/to/LiveTalking && python /path/to/this/repo/scripts/run_amd.py [app.py arguments ...]

It makes the module names app.py imports (`avatars.wav2lip_avatar`, `avatars.musetalk_avatar`,
`avatars.audio_features.mel`, `avatars.audio_features.whisper`; app.py:128-137) resolve to the MI355X plugin modules, then
runs app.py as __main__.  The reference tree stays byte-identical; `avatars.base_avatar`, `registry`, `server/`, `streamout/`
are the reference's own.
"""
import os
import runpy
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ref = os.getcwd()
    if not os.path.exists(os.path.join(ref, "app.py")):
        sys.exit("run_amd.py: start it from the LiveTalking checkout (app.py not found in the current directory)")
    sys.path.insert(0, ref)           # the reference's own packages first: base_avatar, registry, utils, server, streamout
    sys.path.insert(1, REPO)
    import livetalking_amd.hostshim as shim
    if not shim.USING_REFERENCE_HOST:
        sys.exit("run_amd.py: the reference's avatars.base_avatar / registry could not be imported from " + ref)
    import livetalking_amd.avatars.audio_features.mel as mel
    import livetalking_amd.avatars.audio_features.whisper as whisper
    import livetalking_amd.avatars.musetalk_avatar as mt
    import livetalking_amd.avatars.wav2lip_avatar as w2l
    sys.modules["avatars.wav2lip_avatar"] = w2l
    sys.modules["avatars.musetalk_avatar"] = mt
    sys.modules["avatars.audio_features.mel"] = mel
    sys.modules["avatars.audio_features.whisper"] = whisper
    sys.argv = ["app.py"] + sys.argv[1:]
    runpy.run_path(os.path.join(ref, "app.py"), run_name="__main__")


if __name__ == "__main__":
    main()
