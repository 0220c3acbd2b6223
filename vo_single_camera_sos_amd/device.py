"""Thin, checked Python face of the C ABI: torch tensors are used only as device-memory
containers (data_ptr()) and for the stream handle.  Every method validates dtype, device,
contiguity and shape on the host before a kernel may touch the memory."""
import ctypes

import torch

from . import _lib
from ._lib import SosvoError, c_f32, c_p


def _ptr(t):
    return c_p(t.data_ptr())


def _check(t, dtype, name, shape=None, ndim=None):
    if not isinstance(t, torch.Tensor):
        raise SosvoError("%s: expected a torch tensor, got %r" % (name, type(t)))
    if not t.is_cuda:
        raise SosvoError("%s: must live on the GPU" % name)
    if t.dtype != dtype:
        raise SosvoError("%s: dtype %s, expected %s" % (name, t.dtype, dtype))
    if not t.is_contiguous():
        raise SosvoError("%s: must be contiguous" % name)
    if ndim is not None and t.dim() != ndim:
        raise SosvoError("%s: ndim %d, expected %d" % (name, t.dim(), ndim))
    if shape is not None:
        if len(shape) != t.dim() or any(s is not None and s != d for s, d in zip(shape, t.shape)):
            raise SosvoError("%s: shape %s, expected %s" % (name, tuple(t.shape), tuple(shape)))
    return t


class Context(object):
    """One sosvo_ctx bound to (device, stream).  Not shared between host threads."""

    def __init__(self, device=0, stream=None):
        self._lib = _lib.load()
        if not torch.cuda.is_available():
            raise SosvoError("no GPU visible: libsosvo needs an MI355X (gfx950); there is no CPU fallback")
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        if stream is None:
            stream = torch.cuda.current_stream(self.device)
        self.stream = stream
        h = c_p()
        rc = self._lib.sosvo_create(ctypes.byref(h), int(device), c_p(stream.cuda_stream))
        if rc != 0:
            raise SosvoError("sosvo_create failed: %s" % _lib.STATUS_NAMES.get(rc, rc))
        self._h = h

    def close(self):
        if getattr(self, "_h", None) is not None:
            self._lib.sosvo_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _call(self, fn, *args):
        rc = fn(self._h, *args)
        if rc != 0:
            msg = self._lib.sosvo_last_error(self._h)
            raise SosvoError("%s -> %s: %s" % (fn.__name__, _lib.STATUS_NAMES.get(rc, rc),
                                               msg.decode() if msg else ""))

    # ---- plumbing ----------------------------------------------------------------------
    def synchronize(self):
        self._call(self._lib.sosvo_synchronize)

    def timer_start(self):
        self._call(self._lib.sosvo_timer_start)

    def timer_stop(self):
        self._call(self._lib.sosvo_timer_stop)

    def timer_elapsed_ms(self):
        ms = c_f32()
        self._call(self._lib.sosvo_timer_elapsed_ms, ctypes.byref(ms))
        return float(ms.value)

    # ---- K7 ----------------------------------------------------------------------------
    def match_hamming(self, q_desc, t_desc, nq, nt, k=1, keys=None):
        """q_desc [P, Sq, 32] u8, t_desc [P, St, 32] u8, nq/nt [P] i32 -> keys [P, Sq, k] u32
        (stored as int32 bit patterns are NOT used: the tensor dtype is torch.uint32)."""
        _check(q_desc, torch.uint8, "q_desc", (None, None, _lib.DESC_BYTES))
        _check(t_desc, torch.uint8, "t_desc", (q_desc.shape[0], None, _lib.DESC_BYTES))
        P, Sq = q_desc.shape[0], q_desc.shape[1]
        St = t_desc.shape[1]
        _check(nq, torch.int32, "nq", (P,))
        _check(nt, torch.int32, "nt", (P,))
        if keys is None:
            keys = torch.empty((P, Sq, k), dtype=torch.uint32, device=q_desc.device)
        _check(keys, torch.uint32, "keys", (P, Sq, k))
        self._call(self._lib.sosvo_match_hamming, _ptr(q_desc), _ptr(t_desc), _ptr(nq), _ptr(nt),
                   P, Sq, St, int(k), _ptr(keys))
        return keys

    def sort_matches(self, keys, nq, order=None):
        """keys [P, Sq, 1] u32 (1-NN), nq [P] i32 -> order [P, Sq] i32 (query index by rank)."""
        _check(keys, torch.uint32, "keys", (None, None, 1))
        P, Sq = keys.shape[0], keys.shape[1]
        _check(nq, torch.int32, "nq", (P,))
        if order is None:
            order = torch.full((P, Sq), -1, dtype=torch.int32, device=keys.device)
        _check(order, torch.int32, "order", (P, Sq))
        self._call(self._lib.sosvo_sort_matches, _ptr(keys), _ptr(nq), P, Sq, _ptr(order))
        return order
