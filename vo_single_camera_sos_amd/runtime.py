"""Process-wide default libsosvo context (one per host thread, as the C ABI requires)."""
import threading

_local = threading.local()


def default_context(device=None):
    """The calling thread's context on `device` (default: the current torch CUDA device)."""
    import torch
    from .device import Context
    if device is None:
        device = torch.cuda.current_device() if torch.cuda.is_available() else 0
    ctxs = getattr(_local, "ctxs", None)
    if ctxs is None:
        ctxs = _local.ctxs = {}
    if device not in ctxs:
        ctxs[device] = Context(device)
    return ctxs[device]
