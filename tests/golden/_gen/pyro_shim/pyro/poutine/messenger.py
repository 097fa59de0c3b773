from . import runtime


class Messenger(object):
    """Context-manager effect handler (the subset vi.py:492-500 subclasses)."""

    def __init__(self):
        pass

    def __enter__(self):
        runtime._STACK.append(self)
        return self

    def __exit__(self, *exc):
        assert runtime._STACK[-1] is self
        runtime._STACK.pop()
        return False

    def _process_message(self, msg):
        return None

    def _postprocess_message(self, msg):
        return None
