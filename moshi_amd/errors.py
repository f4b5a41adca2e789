"""Exception types of the host side that callers need to tell apart (no GPU, no torch import)."""


class UnknownChannel(ValueError):
    """A session-batcher channel id that names no open channel: closed by a disconnect, or re-opened under a new id by a
    RESTART (mmi_batcher_push_pcm / _close / _pop return MMI_ERR_NO_CHANNEL).  A ValueError like the argument errors, but one a
    model loop can tell apart from them."""
