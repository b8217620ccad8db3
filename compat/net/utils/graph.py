"""Reference import name -> MI355X implementation (see compat/README.md)."""
import speech2affective_gestures_amd.net.utils.graph as _impl
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
