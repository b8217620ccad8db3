"""Reference import name -> MI355X implementation (see compat/README.md)."""
from speech2affective_gestures_amd.net.embedding_space_evaluator import *  # noqa: F401,F403
import speech2affective_gestures_amd.net.embedding_space_evaluator as _impl
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
