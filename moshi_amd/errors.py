"""Exception types of the host side that callers need to tell apart (no GPU, no torch import)."""


class UnknownChannel(ValueError):
    """A session-batcher channel id that names no open channel: closed by a disconnect, or re-opened under a new id by a
    RESTART (mmi_batcher_* return MMI_ERR_INVALID "unknown channel").  A ValueError like every MMI_ERR_INVALID, but one a model
    loop can tell apart from a real argument error."""
