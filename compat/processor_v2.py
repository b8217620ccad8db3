"""Reference import name -> MI355X implementation (see compat/README.md)."""
from speech2affective_gestures_amd.processor_v2 import *  # noqa: F401,F403
from speech2affective_gestures_amd.processor_v2 import Processor, get_epoch_and_loss  # noqa: F401
