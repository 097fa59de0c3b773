"""poutine subset: Messenger base, trace (with param_only), replay.  Fixture generation only."""
from . import runtime  # noqa: F401
from . import messenger  # noqa: F401
from .messenger import Messenger
from .handlers import trace, TraceMessenger, ReplayMessenger, EnumMessenger, Trace  # noqa: F401
