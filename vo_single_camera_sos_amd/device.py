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

    # ---- K8 / K10 / K9 -----------------------------------------------------------------
    def _rig(self, cam, cam_off, cam_rot, P, S):
        """Validate the optional non-central rig description; returns (cam_ptr, off_ptr, rot_ptr, ncam)."""
        if cam is None:
            return c_p(0), c_p(0), c_p(0), 1
        _check(cam, torch.int32, "cam", (P, S))
        _check(cam_off, torch.float64, "cam_off", (None, 3))
        ncam = cam_off.shape[0]
        _check(cam_rot, torch.float64, "cam_rot", (ncam, 3, 3))
        if not 1 <= ncam <= 8:
            raise SosvoError("ncam %d out of range 1..8" % ncam)
        return _ptr(cam), _ptr(cam_off), _ptr(cam_rot), ncam

    def ransac_abs_pose(self, f, p, n, thr, max_iter, seed=0, adaptive=False, cam=None, cam_off=None,
                        cam_rot=None, cam_rot_identity=False, want_counts=False):
        """f, p [P, S, 3] f64, n [P] i32 (+ cam [P, S] i32, cam_off [C,3], cam_rot [C,3,3]) ->
        dict(T [P,3,4], mask [P,S] u8, idx [P,S] i32, n_inliers [P] i32, info [P,4] i32[, counts [P,max_iter]])."""
        _check(f, torch.float64, "f", (None, None, 3))
        P, S = f.shape[0], f.shape[1]
        _check(p, torch.float64, "p", (P, S, 3))
        _check(n, torch.int32, "n", (P,))
        cam_p, off_p, rot_p, ncam = self._rig(cam, cam_off, cam_rot, P, S)
        dev = f.device
        T = torch.empty((P, 3, 4), dtype=torch.float64, device=dev)
        mask = torch.zeros((P, S), dtype=torch.uint8, device=dev)
        idx = torch.full((P, S), -1, dtype=torch.int32, device=dev)
        n_inl = torch.empty((P,), dtype=torch.int32, device=dev)
        info = torch.empty((P, 4), dtype=torch.int32, device=dev)
        counts = torch.empty((P, int(max_iter)), dtype=torch.int32, device=dev) if want_counts else None
        flags = _lib.FLAG_CAM_ROT_IDENTITY if cam_rot_identity else 0
        self._call(self._lib.sosvo_ransac_abs_pose, _ptr(f), _ptr(p), cam_p, off_p, rot_p, ncam, flags,
                   _ptr(n), P, S, float(thr), int(max_iter), 1 if adaptive else 0, int(seed) & (2 ** 64 - 1),
                   _ptr(T), _ptr(mask), _ptr(idx), _ptr(n_inl), _ptr(info),
                   _ptr(counts) if counts is not None else c_p(0))
        out = dict(T=T, mask=mask, idx=idx, n_inliers=n_inl, info=info)
        if want_counts:
            out["counts"] = counts
        return out

    def refine_abs_pose(self, f, p, n, T, idx=None, m=None, cam=None, cam_off=None, cam_rot=None,
                        max_lm_iter=30):
        """In-place LM refinement of T [P,3,4]; returns (T, cost [P] f64, iters [P] i32)."""
        _check(f, torch.float64, "f", (None, None, 3))
        P, S = f.shape[0], f.shape[1]
        _check(p, torch.float64, "p", (P, S, 3))
        _check(n, torch.int32, "n", (P,))
        _check(T, torch.float64, "T", (P, 3, 4))
        cam_p, off_p, rot_p, ncam = self._rig(cam, cam_off, cam_rot, P, S)
        if (idx is None) != (m is None):
            raise SosvoError("idx and m go together")
        if idx is not None:
            _check(idx, torch.int32, "idx", (P, S))
            _check(m, torch.int32, "m", (P,))
        cost = torch.empty((P,), dtype=torch.float64, device=f.device)
        iters = torch.empty((P,), dtype=torch.int32, device=f.device)
        self._call(self._lib.sosvo_refine_abs_pose, _ptr(f), _ptr(p), cam_p, off_p, rot_p, ncam, _ptr(n), P, S,
                   _ptr(idx) if idx is not None else c_p(0), _ptr(m) if m is not None else c_p(0),
                   int(max_lm_iter), _ptr(T), _ptr(cost), _ptr(iters))
        return T, cost, iters
