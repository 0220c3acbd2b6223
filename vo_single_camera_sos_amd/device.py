"""Thin, checked Python face of the C ABI: torch tensors are used only as device-memory
containers (data_ptr()) and for the stream handle.  Every method validates dtype, device,
contiguity and shape on the host before a kernel may touch the memory."""
import ctypes

import torch

from . import _lib
from ._lib import SosvoError, c_f32, c_p

NULL = c_p(0)


def _ptr(t):
    return c_p(t.data_ptr()) if t is not None else NULL


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


def make_rig(pano_top, pano_bot, F_top, F_bot, min_range, max_range, stereo_min_disp=1.0, stereo_max_hdiff=2.5,
             f2f_max_hdiff=-1.0, pct_good_matches=1.0):
    """Host-side sosvo_rig.  pano_* = (cols, rows, pixel_size, cyl_height_max)."""
    r = _lib.Rig()
    for k in range(4):
        r.pano_top[k] = float(pano_top[k])
        r.pano_bot[k] = float(pano_bot[k])
    for k in range(3):
        r.F_top[k] = float(F_top[k])
        r.F_bot[k] = float(F_bot[k])
    r.min_range, r.max_range = float(min_range), float(max_range)
    r.stereo_min_disp, r.stereo_max_hdiff = float(stereo_min_disp), float(stereo_max_hdiff)
    r.f2f_max_hdiff, r.pct_good_matches = float(f2f_max_hdiff), float(pct_good_matches)
    return r


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
    def set_hint_shared_device(self, on=True):
        """sosvo_set_hint(SOSVO_HINT_SHARED_DEVICE): other contexts' kernels run beside this one's (scheduling only)."""
        self._call(self._lib.sosvo_set_hint, _lib.HINT_SHARED_DEVICE, 1 if on else 0)

    def set_hint_score_fp64_only(self, on=True):
        """sosvo_set_hint(SOSVO_HINT_SCORE_FP64_ONLY): RANSAC scoring without its single-precision first tier (same counts;
        for A/B measurements and the cross-check of the two forms)."""
        self._call(self._lib.sosvo_set_hint, _lib.HINT_SCORE_FP64_ONLY, 1 if on else 0)

    def set_stream(self, stream):
        """Enqueue on `stream` (a torch.cuda.Stream) from now on, e.g. the capture stream of a HIP graph."""
        self._call(self._lib.sosvo_set_stream, c_p(stream.cuda_stream))
        self.stream = stream

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

    def debug_fill_scratch(self, byte=0xFF):
        """Test hook: overwrite the library's internal scratch memory of this context (results must not change)."""
        self._call(self._lib.sosvo_debug_fill_scratch, int(byte))

    def profile_enable(self, on=True):
        """Start (and clear) / stop the per-kernel HIP-event record."""
        self._call(self._lib.sosvo_profile_enable, 1 if on else 0)

    def profile_read(self):
        """-> list of (kernel label, milliseconds) for every launch since profile_enable(True)."""
        n = self._lib.sosvo_profile_count(self._h)
        buf = ctypes.create_string_buffer(128)
        ms = c_f32()
        out = []
        for i in range(n):
            self._call(self._lib.sosvo_profile_get, i, buf, 128, ctypes.byref(ms))
            out.append((buf.value.decode(), float(ms.value)))
        return out

    # ---- K1 / K2 / K3 ------------------------------------------------------------------
    def unwrap(self, omni, masks, map_x, map_y, pano=None):
        """omni [F,H,W,3] u8, masks [2,H,W] u8 or None, map_x/map_y [2,rows,cols] f32 ->
        pano [2,F,rows,cols,3] u8 (view-major)."""
        _check(omni, torch.uint8, "omni", (None, None, None, 3))
        F, H, W = omni.shape[0], omni.shape[1], omni.shape[2]
        if masks is not None:
            _check(masks, torch.uint8, "masks", (2, H, W))
        _check(map_x, torch.float32, "map_x", (2, None, None))
        rows, cols = map_x.shape[1], map_x.shape[2]
        _check(map_y, torch.float32, "map_y", (2, rows, cols))
        if pano is None:
            pano = torch.empty((2, F, rows, cols, 3), dtype=torch.uint8, device=omni.device)
        _check(pano, torch.uint8, "pano", (2, F, rows, cols, 3))
        self._call(self._lib.sosvo_unwrap, _ptr(omni), _ptr(masks), _ptr(map_x), _ptr(map_y), F, H, W, rows, cols,
                   _ptr(pano))
        return pano

    def unwrap_prepare(self, masks, map_x, map_y, omni_shape):
        """Model constants -> packed unwrap table [2, rows, cols, 2] u32 (once per model)."""
        H, W = omni_shape
        if masks is not None:
            _check(masks, torch.uint8, "masks", (2, H, W))
        _check(map_x, torch.float32, "map_x", (2, None, None))
        rows, cols = map_x.shape[1], map_x.shape[2]
        _check(map_y, torch.float32, "map_y", (2, rows, cols))
        table = torch.empty((2, rows, cols, 2), dtype=torch.uint32, device=map_x.device)
        self._call(self._lib.sosvo_unwrap_prepare, _ptr(masks), _ptr(map_x), _ptr(map_y), H, W, rows, cols, _ptr(table))
        return table

    def unwrap_table(self, omni, table, pano=None):
        """omni [F,H,W,3] u8, table from unwrap_prepare -> pano [2,F,rows,cols,3] u8 (view-major)."""
        _check(omni, torch.uint8, "omni", (None, None, None, 3))
        F, H, W = omni.shape[0], omni.shape[1], omni.shape[2]
        _check(table, torch.uint32, "table", (2, None, None, 2))
        rows, cols = table.shape[1], table.shape[2]
        if pano is None:
            pano = torch.empty((2, F, rows, cols, 3), dtype=torch.uint8, device=omni.device)
        _check(pano, torch.uint8, "pano", (2, F, rows, cols, 3))
        self._call(self._lib.sosvo_unwrap_table, _ptr(omni), _ptr(table), F, H, W, rows, cols, _ptr(pano))
        return pano

    def median_gray(self, img, ksize, gray=None):
        """img [..., rows, cols, 3] u8 -> gray [..., rows, cols] u8 (k x k median per channel, then BGR2GRAY)."""
        _check(img, torch.uint8, "img")
        if img.dim() < 3 or img.shape[-1] != 3:
            raise SosvoError("img: expected [..., rows, cols, 3]")
        rows, cols = img.shape[-3], img.shape[-2]
        nimg = int(img.numel() // (rows * cols * 3))
        if gray is None:
            gray = torch.empty(tuple(img.shape[:-1]), dtype=torch.uint8, device=img.device)
        _check(gray, torch.uint8, "gray", tuple(img.shape[:-1]))
        self._call(self._lib.sosvo_median_gray, _ptr(img), nimg, rows, cols, int(ksize), _ptr(gray))
        return gray

    def gray_rows_needed(self, mask_bits, nmask, edge, pattern, cos_a, sin_a):
        """Rows of the gray panoramas that GFT on the azimuthal masks + ORB.compute can reach: mask_bits
        [nsets,rows,cols] u32 -> row_range [nsets,2] i32 on the device (first row, last row + 1).  Model constant."""
        _check(mask_bits, torch.uint32, "mask_bits", (None, None, None))
        _check(pattern, torch.int8, "pattern")
        nsets, rows, cols = mask_bits.shape
        out = torch.zeros((nsets, 2), dtype=torch.int32, device=mask_bits.device)
        self._call(self._lib.sosvo_gray_rows_needed, _ptr(mask_bits), nsets, rows, cols, int(nmask), int(edge), _ptr(pattern),
                   float(cos_a), float(sin_a), _ptr(out))
        return out

    def unwrap_median_gray(self, omni, table, ksize, gray=None, row_range=None):
        """K1 + K2 + K3 fused: omni [F,H,W,3] u8, table from unwrap_prepare -> gray [2F,rows,cols] u8 (view-major);
        the colour panoramas are never materialised.  Same result as unwrap_table + median_gray.  row_range
        ([2,2] i32 from gray_rows_needed): only those rows of each view are computed and written."""
        _check(omni, torch.uint8, "omni", (None, None, None, 3))
        F, H, W = omni.shape[0], omni.shape[1], omni.shape[2]
        _check(table, torch.uint32, "table", (2, None, None, 2))
        rows, cols = table.shape[1], table.shape[2]
        if gray is None:
            gray = torch.empty((2 * F, rows, cols), dtype=torch.uint8, device=omni.device)
        _check(gray, torch.uint8, "gray", (2 * F, rows, cols))
        if row_range is None:
            self._call(self._lib.sosvo_unwrap_median_gray, _ptr(omni), _ptr(table), F, H, W, rows, cols, int(ksize), _ptr(gray))
        else:
            _check(row_range, torch.int32, "row_range", (2, 2))
            self._call(self._lib.sosvo_unwrap_median_gray_rows, _ptr(omni), _ptr(table), F, H, W, rows, cols, int(ksize),
                       _ptr(row_range), _ptr(gray))
        return gray

    def detect_agast(self, gray, mask_bits, images_per_maskset, nmask, cap, threshold=10, kp=None, n=None, status=None):
        """AGAST (OAST 9/16) + its block-maximum NMS per azimuthal mask, raster order; arguments as detect_fast."""
        return self.detect_fast(gray, mask_bits, images_per_maskset, nmask, cap, threshold, kp, n, status, _agast=True)

    def detect_fast(self, gray, mask_bits, images_per_maskset, nmask, cap, threshold=10, kp=None, n=None, status=None,
                    _agast=False):
        """FAST-9/16 + NMS per azimuthal mask, raster order: gray [NI,rows,cols] u8, mask_bits [nsets,rows,cols] u32
        -> kp [NI*nmask, cap, 2] f32, n [NI*nmask] i32, status [NI*nmask] i32."""
        _check(gray, torch.uint8, "gray", ndim=3)
        NI, rows, cols = gray.shape
        _check(mask_bits, torch.uint32, "mask_bits", (None, rows, cols))
        if mask_bits.shape[0] * images_per_maskset < NI:
            raise SosvoError("mask_bits has too few sets for %d images" % NI)
        P, dev = NI * nmask, gray.device
        if kp is None:
            kp = torch.zeros((P, cap, 2), dtype=torch.float32, device=dev)
        if n is None:
            n = torch.zeros((P,), dtype=torch.int32, device=dev)
        if status is None:
            status = torch.zeros((P,), dtype=torch.int32, device=dev)
        _check(kp, torch.float32, "kp", (P, cap, 2))
        _check(n, torch.int32, "n", (P,))
        _check(status, torch.int32, "status", (P,))
        self._call(self._lib.sosvo_detect_agast if _agast else self._lib.sosvo_detect_fast, _ptr(gray), _ptr(mask_bits), NI,
                   int(images_per_maskset), rows, cols, int(nmask), int(threshold), int(cap), _ptr(kp), _ptr(n), _ptr(status))
        return kp, n, status

    # ---- K4 / K6 -----------------------------------------------------------------------
    def detect_gft(self, gray, mask_bits, images_per_maskset, nmask, cap, quality=0.01, min_distance=5.0,
                   max_corners=1000, kp=None, n=None, status=None):
        """gray [NI,rows,cols] u8, mask_bits [nsets,rows,cols] u32 (as int32 bit patterns are not accepted: use
        torch.uint32) -> kp [NI*nmask, cap, 2] f32, n [NI*nmask] i32, status [NI*nmask] i32."""
        _check(gray, torch.uint8, "gray", ndim=3)
        NI, rows, cols = gray.shape
        _check(mask_bits, torch.uint32, "mask_bits", (None, rows, cols))
        if mask_bits.shape[0] * images_per_maskset < NI:
            raise SosvoError("mask_bits has too few sets for %d images" % NI)
        P = NI * nmask
        dev = gray.device
        if kp is None:
            kp = torch.zeros((P, cap, 2), dtype=torch.float32, device=dev)
        if n is None:
            n = torch.zeros((P,), dtype=torch.int32, device=dev)
        if status is None:
            status = torch.zeros((P,), dtype=torch.int32, device=dev)
        _check(kp, torch.float32, "kp", (P, cap, 2))
        _check(n, torch.int32, "n", (P,))
        _check(status, torch.int32, "status", (P,))
        self._call(self._lib.sosvo_detect_gft, _ptr(gray), _ptr(mask_bits), NI, int(images_per_maskset), rows, cols,
                   int(nmask), float(quality), float(min_distance), int(max_corners), int(cap), _ptr(kp), _ptr(n),
                   _ptr(status))
        return kp, n, status

    def describe_orb(self, gray, kp, n, nmask, pattern, cos_a, sin_a, edge=31, desc=None, row_range=None):
        """Compacts kp/n in place (border rule) and returns desc [NI*nmask, cap, 32] u8.  row_range (int32 [2,2] on the
        device, from gray_rows_needed): blur only the rows a descriptor can read."""
        _check(gray, torch.uint8, "gray", ndim=3)
        NI, rows, cols = gray.shape
        P = NI * nmask
        _check(kp, torch.float32, "kp", (P, None, 2))
        cap = kp.shape[1]
        _check(n, torch.int32, "n", (P,))
        _check(pattern, torch.int8, "pattern", (512, 2))
        if desc is None:
            desc = torch.zeros((P, cap, 32), dtype=torch.uint8, device=gray.device)
        _check(desc, torch.uint8, "desc", (P, cap, 32))
        if row_range is not None:
            _check(row_range, torch.int32, "row_range", (2, 2))
        self._call(self._lib.sosvo_describe_orb_rows, _ptr(gray), NI, rows, cols, int(nmask), cap, _ptr(kp), _ptr(n),
                   float(cos_a), float(sin_a), _ptr(pattern), int(edge), _ptr(row_range), _ptr(desc))
        return desc

    # ---- K5 / K6' ----------------------------------------------------------------------
    def orb_pyramid_pixels(self, rows, cols):
        return int(self._lib.sosvo_orb_pyramid_pixels(int(rows), int(cols)))

    def orb_mask_pyramid(self, mask_bits, nmask):
        """mask_bits [nsets, rows, cols] u32 -> mask_pyr [nsets, pyramid_pixels] u32 (once per model)."""
        _check(mask_bits, torch.uint32, "mask_bits", ndim=3)
        nsets, rows, cols = mask_bits.shape
        out = torch.zeros((nsets, self.orb_pyramid_pixels(rows, cols)), dtype=torch.uint32, device=mask_bits.device)
        self._call(self._lib.sosvo_orb_mask_pyramid, _ptr(mask_bits), nsets, rows, cols, int(nmask), _ptr(out))
        return out

    def detect_orb(self, gray, mask_pyr, images_per_maskset, nmask, nfeatures, cap, kp4=None, resp=None, n=None):
        """gray [NI,rows,cols] u8 -> kp4 [NI*nmask, cap, 4] f32 (x, y, angle_deg, level), resp [NI*nmask, cap] f32,
        n [NI*nmask] i32."""
        _check(gray, torch.uint8, "gray", ndim=3)
        NI, rows, cols = gray.shape
        _check(mask_pyr, torch.uint32, "mask_pyr", (None, self.orb_pyramid_pixels(rows, cols)))
        if mask_pyr.shape[0] * images_per_maskset < NI:
            raise SosvoError("mask_pyr has too few sets for %d images" % NI)
        P = NI * nmask
        dev = gray.device
        if kp4 is None:
            kp4 = torch.zeros((P, cap, 4), dtype=torch.float32, device=dev)
        if resp is None:
            resp = torch.zeros((P, cap), dtype=torch.float32, device=dev)
        if n is None:
            n = torch.zeros((P,), dtype=torch.int32, device=dev)
        _check(kp4, torch.float32, "kp4", (P, cap, 4))
        _check(resp, torch.float32, "resp", (P, cap))
        _check(n, torch.int32, "n", (P,))
        self._call(self._lib.sosvo_detect_orb, _ptr(gray), _ptr(mask_pyr), NI, int(images_per_maskset), rows, cols,
                   int(nmask), int(nfeatures), int(cap), _ptr(kp4), _ptr(resp), _ptr(n))
        return kp4, resp, n

    def describe_orb_levels(self, gray, kp4, n, nmask, pattern, desc=None, kp_xy=None):
        """Compacts kp4 / n in place (31-px border rule); returns (desc [P,cap,32] u8, kp_xy [P,cap,2] f32)."""
        _check(gray, torch.uint8, "gray", ndim=3)
        NI, rows, cols = gray.shape
        P = NI * nmask
        _check(kp4, torch.float32, "kp4", (P, None, 4))
        cap = kp4.shape[1]
        _check(n, torch.int32, "n", (P,))
        _check(pattern, torch.int8, "pattern", (512, 2))
        if desc is None:
            desc = torch.zeros((P, cap, 32), dtype=torch.uint8, device=gray.device)
        if kp_xy is None:
            kp_xy = torch.zeros((P, cap, 2), dtype=torch.float32, device=gray.device)
        _check(desc, torch.uint8, "desc", (P, cap, 32))
        _check(kp_xy, torch.float32, "kp_xy", (P, cap, 2))
        self._call(self._lib.sosvo_describe_orb_levels, _ptr(gray), NI, rows, cols, int(nmask), cap, _ptr(kp4), _ptr(n),
                   _ptr(pattern), _ptr(desc), _ptr(kp_xy))
        return desc, kp_xy

    def detect_describe_orb(self, gray, mask_pyr, images_per_maskset, nmask, nfeatures, pattern, kp4, resp, n, desc,
                            kp_xy=None):
        """ORB detect + compute in one call on ONE shared pyramid (sosvo_detect_describe_orb): the outputs of detect_orb
        followed by describe_orb_levels (kp4 / n compacted by the border rule), all buffers caller-provided."""
        _check(gray, torch.uint8, "gray", ndim=3)
        NI, rows, cols = gray.shape
        P = NI * nmask
        _check(mask_pyr, torch.uint32, "mask_pyr", (None, self.orb_pyramid_pixels(rows, cols)))
        _check(kp4, torch.float32, "kp4", (P, None, 4))
        cap = kp4.shape[1]
        _check(resp, torch.float32, "resp", (P, cap))
        _check(n, torch.int32, "n", (P,))
        _check(pattern, torch.int8, "pattern", (512, 2))
        _check(desc, torch.uint8, "desc", (P, cap, 32))
        if kp_xy is not None:
            _check(kp_xy, torch.float32, "kp_xy", (P, cap, 2))
        self._call(self._lib.sosvo_detect_describe_orb, _ptr(gray), _ptr(mask_pyr), NI, int(images_per_maskset), rows, cols,
                   int(nmask), int(nfeatures), cap, _ptr(kp4), _ptr(resp), _ptr(n), _ptr(pattern), _ptr(desc),
                   _ptr(kp_xy) if kp_xy is not None else None)
        return kp4, resp, n, desc, kp_xy

    # ---- K7 ----------------------------------------------------------------------------
    def match_hamming(self, q_desc, t_desc, nq, nt, k=1, keys=None, q_slot=None, t_slot=None):
        """q_desc [Bq, Sq, 32] u8, t_desc [Bt, St, 32] u8, nq [Bq] / nt [Bt] i32 -> keys [P, Sq, k] u32.
        Without slots P = Bq = Bt; with q_slot / t_slot [P] i32, problem p uses blocks q_slot[p] / t_slot[p]."""
        _check(q_desc, torch.uint8, "q_desc", (None, None, _lib.DESC_BYTES))
        _check(t_desc, torch.uint8, "t_desc", (None, None, _lib.DESC_BYTES))
        Bq, Sq = q_desc.shape[0], q_desc.shape[1]
        Bt, St = t_desc.shape[0], t_desc.shape[1]
        _check(nq, torch.int32, "nq", (Bq,))
        _check(nt, torch.int32, "nt", (Bt,))
        if (q_slot is None) != (t_slot is None):
            raise SosvoError("q_slot and t_slot go together")
        if q_slot is not None:
            P = q_slot.shape[0]
            _check(q_slot, torch.int32, "q_slot", (P,))
            _check(t_slot, torch.int32, "t_slot", (P,))
        else:
            if Bq != Bt:
                raise SosvoError("q_desc and t_desc must have the same number of blocks without slots")
            P = Bq
        if keys is None:
            keys = torch.empty((P, Sq, k), dtype=torch.uint32, device=q_desc.device)
        _check(keys, torch.uint32, "keys", (P, Sq, k))
        self._call(self._lib.sosvo_match_hamming, _ptr(q_desc), _ptr(t_desc), _ptr(nq), _ptr(nt), _ptr(q_slot),
                   _ptr(t_slot), P, Sq, St, int(k), _ptr(keys))
        return keys

    def match_l2(self, q_desc, t_desc, nq, nt, k=1):
        """Float descriptors: q_desc [P, Sq, dim] f32, t_desc [P, St, dim] f32 -> keys [P, Sq, k] int64 holding the u64
        (float32 distance bits << 32 | train index); all ones (-1) if absent."""
        _check(q_desc, torch.float32, "q_desc", (None, None, None))
        P, Sq, dim = q_desc.shape
        _check(t_desc, torch.float32, "t_desc", (P, None, dim))
        St = t_desc.shape[1]
        _check(nq, torch.int32, "nq", (P,))
        _check(nt, torch.int32, "nt", (P,))
        keys = torch.empty((P, Sq, int(k)), dtype=torch.int64, device=q_desc.device)
        self._call(self._lib.sosvo_match_l2, _ptr(q_desc), _ptr(t_desc), _ptr(nq), _ptr(nt), P, Sq, St, int(dim), int(k),
                   _ptr(keys))
        return keys

    def match_radius(self, q_desc, t_desc, nq, nt, max_distance, cap, q_slot=None, t_slot=None):
        """radiusMatch: q_desc [Bq, Sq, 32] u8, t_desc [Bt, St, 32] u8 -> (keys [P, Sq, cap] u32 ascending,
        KEY_NONE padded; counts [P, Sq] i32 = matches within max_distance, may exceed cap)."""
        _check(q_desc, torch.uint8, "q_desc", (None, None, _lib.DESC_BYTES))
        _check(t_desc, torch.uint8, "t_desc", (None, None, _lib.DESC_BYTES))
        Bq, Sq = q_desc.shape[0], q_desc.shape[1]
        Bt, St = t_desc.shape[0], t_desc.shape[1]
        _check(nq, torch.int32, "nq", (Bq,))
        _check(nt, torch.int32, "nt", (Bt,))
        if (q_slot is None) != (t_slot is None):
            raise SosvoError("q_slot and t_slot go together")
        if q_slot is not None:
            P = q_slot.shape[0]
            _check(q_slot, torch.int32, "q_slot", (P,))
            _check(t_slot, torch.int32, "t_slot", (P,))
        else:
            if Bq != Bt:
                raise SosvoError("q_desc and t_desc must have the same number of blocks without slots")
            P = Bq
        keys = torch.empty((P, Sq, int(cap)), dtype=torch.uint32, device=q_desc.device)
        counts = torch.zeros((P, Sq), dtype=torch.int32, device=q_desc.device)
        self._call(self._lib.sosvo_match_radius, _ptr(q_desc), _ptr(t_desc), _ptr(nq), _ptr(nt), _ptr(q_slot), _ptr(t_slot),
                   P, Sq, St, int(max_distance), int(cap), _ptr(keys), _ptr(counts))
        return keys, counts

    def sort_matches(self, keys, nq, order=None, q_slot=None):
        """keys [P, Sq, 1] u32 (1-NN), nq i32 (per block) -> order [P, Sq] i32 (query index by rank)."""
        _check(keys, torch.uint32, "keys", (None, None, 1))
        P, Sq = keys.shape[0], keys.shape[1]
        _check(nq, torch.int32, "nq", ndim=1)
        if q_slot is not None:
            _check(q_slot, torch.int32, "q_slot", (P,))
        elif nq.shape[0] != P:
            raise SosvoError("nq must have one entry per problem without q_slot")
        if order is None:
            order = torch.full((P, Sq), -1, dtype=torch.int32, device=keys.device)
        _check(order, torch.int32, "order", (P, Sq))
        self._call(self._lib.sosvo_sort_matches, _ptr(keys), _ptr(nq), _ptr(q_slot), P, Sq, _ptr(order))
        return order

    # ---- K8 / K10 / K9 -----------------------------------------------------------------
    def _rig_cams(self, cam, cam_off, cam_rot, P, S):
        """Validate the optional non-central rig description; returns (cam_ptr, off_ptr, rot_ptr, ncam)."""
        if cam is None:
            return NULL, NULL, NULL, 1
        _check(cam, torch.int32, "cam", (P, S))
        _check(cam_off, torch.float64, "cam_off", (None, 3))
        ncam = cam_off.shape[0]
        _check(cam_rot, torch.float64, "cam_rot", (ncam, 3, 3))
        if not 1 <= ncam <= 8:
            raise SosvoError("ncam %d out of range 1..8" % ncam)
        return _ptr(cam), _ptr(cam_off), _ptr(cam_rot), ncam

    def ransac_abs_pose(self, f, p, n, thr, max_iter, seed=0, adaptive=False, cam=None, cam_off=None,
                        cam_rot=None, cam_rot_identity=False, want_counts=False, out=None, epnp=False, gp3p=False, twopt=False):
        """f, p [P, S, 3] f64, n [P] i32 (+ cam [P, S] i32, cam_off [C,3], cam_rot [C,3,3]) ->
        dict(T [P,3,4], mask [P,S] u8, idx [P,S] i32, n_inliers [P] i32, info [P,4] i32[, counts [P,max_iter]]).
        `out` may carry preallocated result tensors (same keys) to avoid allocations in a hot loop."""
        _check(f, torch.float64, "f", (None, None, 3))
        P, S = f.shape[0], f.shape[1]
        _check(p, torch.float64, "p", (P, S, 3))
        _check(n, torch.int32, "n", (P,))
        cam_p, off_p, rot_p, ncam = self._rig_cams(cam, cam_off, cam_rot, P, S)
        dev = f.device
        if out is None:
            out = dict(T=torch.empty((P, 3, 4), dtype=torch.float64, device=dev),
                       mask=torch.zeros((P, S), dtype=torch.uint8, device=dev),
                       idx=torch.full((P, S), -1, dtype=torch.int32, device=dev),
                       n_inliers=torch.empty((P,), dtype=torch.int32, device=dev),
                       info=torch.empty((P, 4), dtype=torch.int32, device=dev))
            if want_counts:
                out["counts"] = torch.empty((P, int(max_iter)), dtype=torch.int32, device=dev)
        _check(out["T"], torch.float64, "T", (P, 3, 4))
        _check(out["mask"], torch.uint8, "mask", (P, S))
        _check(out["idx"], torch.int32, "idx", (P, S))
        _check(out["n_inliers"], torch.int32, "n_inliers", (P,))
        _check(out["info"], torch.int32, "info", (P, 4))
        counts = out.get("counts")
        if counts is not None:
            _check(counts, torch.int32, "counts", (P, int(max_iter)))
        flags = (_lib.FLAG_CAM_ROT_IDENTITY if cam_rot_identity else 0) | (_lib.FLAG_EPNP if epnp else 0) | \
            (_lib.FLAG_GP3P if gp3p else 0) | (_lib.FLAG_TWOPT if twopt else 0)
        self._call(self._lib.sosvo_ransac_abs_pose, _ptr(f), _ptr(p), cam_p, off_p, rot_p, ncam, flags,
                   _ptr(n), P, S, float(thr), int(max_iter), 1 if adaptive else 0, int(seed) & (2 ** 64 - 1),
                   _ptr(out["T"]), _ptr(out["mask"]), _ptr(out["idx"]), _ptr(out["n_inliers"]), _ptr(out["info"]),
                   _ptr(counts))
        return out

    def ransac_rel_pose(self, f1, f2, n, thr, max_iter, algorithm=_lib.REL_EIGHTPT, seed=0, adaptive=False, want_counts=False):
        """f1, f2 [P, S, 3] f64 (bearings of the same features from viewpoints 1 and 2), n [P] i32 ->
        dict(T [P,3,4] (pose of viewpoint 2 in frame 1, |t| = 1), mask, idx, n_inliers, info[, counts])."""
        _check(f1, torch.float64, "f1", (None, None, 3))
        P, S = f1.shape[0], f1.shape[1]
        _check(f2, torch.float64, "f2", (P, S, 3))
        _check(n, torch.int32, "n", (P,))
        dev = f1.device
        out = dict(T=torch.empty((P, 3, 4), dtype=torch.float64, device=dev),
                   mask=torch.zeros((P, S), dtype=torch.uint8, device=dev),
                   idx=torch.full((P, S), -1, dtype=torch.int32, device=dev),
                   n_inliers=torch.empty((P,), dtype=torch.int32, device=dev),
                   info=torch.empty((P, 4), dtype=torch.int32, device=dev))
        if want_counts:
            out["counts"] = torch.empty((P, int(max_iter)), dtype=torch.int32, device=dev)
        self._call(self._lib.sosvo_ransac_rel_pose, _ptr(f1), _ptr(f2), _ptr(n), P, S, int(algorithm), float(thr), int(max_iter),
                   1 if adaptive else 0, int(seed) & (2 ** 64 - 1), _ptr(out["T"]), _ptr(out["mask"]), _ptr(out["idx"]),
                   _ptr(out["n_inliers"]), _ptr(out["info"]), _ptr(out.get("counts")))
        return out

    def refine_abs_pose(self, f, p, n, T, idx=None, m=None, cam=None, cam_off=None, cam_rot=None,
                        max_lm_iter=30, cost=None, iters=None):
        """In-place LM refinement of T [P,3,4]; returns (T, cost [P] f64, iters [P] i32)."""
        _check(f, torch.float64, "f", (None, None, 3))
        P, S = f.shape[0], f.shape[1]
        _check(p, torch.float64, "p", (P, S, 3))
        _check(n, torch.int32, "n", (P,))
        _check(T, torch.float64, "T", (P, 3, 4))
        cam_p, off_p, rot_p, ncam = self._rig_cams(cam, cam_off, cam_rot, P, S)
        if (idx is None) != (m is None):
            raise SosvoError("idx and m go together")
        if idx is not None:
            _check(idx, torch.int32, "idx", (P, S))
            _check(m, torch.int32, "m", (P,))
        if cost is None:
            cost = torch.empty((P,), dtype=torch.float64, device=f.device)
        if iters is None:
            iters = torch.empty((P,), dtype=torch.int32, device=f.device)
        _check(cost, torch.float64, "cost", (P,))
        _check(iters, torch.int32, "iters", (P,))
        self._call(self._lib.sosvo_refine_abs_pose, _ptr(f), _ptr(p), cam_p, off_p, rot_p, ncam, _ptr(n), P, S,
                   _ptr(idx), _ptr(m), int(max_lm_iter), _ptr(T), _ptr(cost), _ptr(iters))
        return T, cost, iters

    # ---- geometry ----------------------------------------------------------------------
    def pano_to_bearing(self, uv, cols, rows, pixel_size, cyl_height_max):
        """uv [n,2] f64 -> (az [n], el [n], bearing [n,3]) f64."""
        _check(uv, torch.float64, "uv", (None, 2))
        n = uv.shape[0]
        az = torch.empty((n,), dtype=torch.float64, device=uv.device)
        el = torch.empty((n,), dtype=torch.float64, device=uv.device)
        b = torch.empty((n, 3), dtype=torch.float64, device=uv.device)
        self._call(self._lib.sosvo_pano_to_bearing, _ptr(uv), n, float(cols), float(rows), float(pixel_size),
                   float(cyl_height_max), _ptr(az), _ptr(el), _ptr(b))
        return az, el, b

    def triangulate_midpoint(self, az_top, el_top, az_bot, el_bot, F_top, F_bot):
        n = az_top.shape[0]
        for name, t in (("az_top", az_top), ("el_top", el_top), ("az_bot", az_bot), ("el_bot", el_bot)):
            _check(t, torch.float64, name, (n,))
        X = torch.empty((n, 3), dtype=torch.float64, device=az_top.device)
        f1 = (ctypes.c_double * 3)(*[float(v) for v in F_top])
        f2 = (ctypes.c_double * 3)(*[float(v) for v in F_bot])
        self._call(self._lib.sosvo_triangulate_midpoint, _ptr(az_top), _ptr(el_top), _ptr(az_bot), _ptr(el_bot), n,
                   ctypes.cast(f1, c_p), ctypes.cast(f2, c_p), _ptr(X))
        return X

    def triangulate2(self, b1, b2, t12, R12):
        """b1, b2 [n,3] f64 unit bearings in frames 1 / 2, t12 (3), R12 (3x3) host -> X [n,3] in frame 1."""
        n = b1.shape[0]
        _check(b1, torch.float64, "b1", (n, 3))
        _check(b2, torch.float64, "b2", (n, 3))
        X = torch.empty((n, 3), dtype=torch.float64, device=b1.device)
        t = (ctypes.c_double * 3)(*[float(v) for v in t12])
        R = (ctypes.c_double * 9)(*[float(v) for row in R12 for v in row])
        self._call(self._lib.sosvo_triangulate2, _ptr(b1), _ptr(b2), n, ctypes.cast(t, c_p), ctypes.cast(R, c_p), _ptr(X))
        return X

    def range_filter(self, X, min_range, max_range):
        _check(X, torch.float64, "X", (None, 3))
        ok = torch.empty((X.shape[0],), dtype=torch.uint8, device=X.device)
        self._call(self._lib.sosvo_range_filter, _ptr(X), X.shape[0], float(min_range), float(max_range), _ptr(ok))
        return ok

    def rgbd_backproject(self, depth, u, v, fx, fy, cx, cy, focal_length_m, depth_is_Z):
        _check(depth, torch.float32, "depth", ndim=2)
        n = u.shape[0]
        _check(u, torch.int32, "u", (n,))
        _check(v, torch.int32, "v", (n,))
        xyz = torch.empty((n, 3), dtype=torch.float64, device=depth.device)
        b = torch.empty((n, 3), dtype=torch.float64, device=depth.device)
        self._call(self._lib.sosvo_rgbd_backproject, _ptr(depth), depth.shape[0], depth.shape[1], _ptr(u), _ptr(v), n,
                   float(fx), float(fy), float(cx), float(cy), float(focal_length_m), 1 if depth_is_Z else 0,
                   _ptr(xyz), _ptr(b))
        return xyz, b

    def stereo_assemble(self, rig, kp_top, kp_bot, desc_top, desc_bot, n_top, n_bot, keys, order, nframes, nmask,
                        out_cap, out=None):
        """Per-bucket stereo matches -> per-frame correspondences (see include/sosvo.h).
        kp_* [F*NM, cap, 2] f32, desc_* [F*NM, cap, 32] u8, n_* [F*NM] i32, keys [F*NM, cap, 1] u32,
        order [F*NM, cap] i32 -> dict(m_top, m_bot [F,out_cap,2] f32, d_top, d_bot [F,out_cap,32] u8,
        X, b_top, b_bot [F,out_cap,3] f64, M [F] i32, n_cand [F] i32)."""
        P = nframes * nmask
        _check(kp_top, torch.float32, "kp_top", (P, None, 2))
        cap = kp_top.shape[1]
        _check(kp_bot, torch.float32, "kp_bot", (P, cap, 2))
        _check(desc_top, torch.uint8, "desc_top", (P, cap, 32))
        _check(desc_bot, torch.uint8, "desc_bot", (P, cap, 32))
        _check(n_top, torch.int32, "n_top", (P,))
        _check(n_bot, torch.int32, "n_bot", (P,))
        _check(keys, torch.uint32, "keys", (P, cap, 1))
        _check(order, torch.int32, "order", (P, cap))
        dev = kp_top.device
        F = nframes
        if out is None:
            out = dict(m_top=torch.zeros((F, out_cap, 2), dtype=torch.float32, device=dev),
                       m_bot=torch.zeros((F, out_cap, 2), dtype=torch.float32, device=dev),
                       d_top=torch.zeros((F, out_cap, 32), dtype=torch.uint8, device=dev),
                       d_bot=torch.zeros((F, out_cap, 32), dtype=torch.uint8, device=dev),
                       X=torch.zeros((F, out_cap, 3), dtype=torch.float64, device=dev),
                       b_top=torch.zeros((F, out_cap, 3), dtype=torch.float64, device=dev),
                       b_bot=torch.zeros((F, out_cap, 3), dtype=torch.float64, device=dev),
                       M=torch.zeros((F,), dtype=torch.int32, device=dev),
                       n_cand=torch.zeros((F,), dtype=torch.int32, device=dev))
        _check(out["m_top"], torch.float32, "m_top", (F, out_cap, 2))
        _check(out["m_bot"], torch.float32, "m_bot", (F, out_cap, 2))
        _check(out["d_top"], torch.uint8, "d_top", (F, out_cap, 32))
        _check(out["d_bot"], torch.uint8, "d_bot", (F, out_cap, 32))
        for k in ("X", "b_top", "b_bot"):
            _check(out[k], torch.float64, k, (F, out_cap, 3))
        _check(out["M"], torch.int32, "M", (F,))
        _check(out["n_cand"], torch.int32, "n_cand", (F,))
        self._call(self._lib.sosvo_stereo_assemble, ctypes.cast(ctypes.pointer(rig), c_p), _ptr(kp_top), _ptr(kp_bot),
                   _ptr(desc_top), _ptr(desc_bot), _ptr(n_top), _ptr(n_bot), _ptr(keys), _ptr(order), F, nmask, cap,
                   out_cap, _ptr(out["m_top"]), _ptr(out["m_bot"]), _ptr(out["d_top"]), _ptr(out["d_bot"]),
                   _ptr(out["X"]), _ptr(out["b_top"]), _ptr(out["b_bot"]), _ptr(out["M"]), _ptr(out["n_cand"]))
        return out

    # ---- RGB-D variant ---------------------------------------------------------------------
    def rgbd_assemble(self, cam, kp, desc, n, depth, out_cap, out=None):
        """RGBDFrame.establish_keypoints after detection, batched.  cam: _lib.RgbdCam; kp [F,cap,2] f32,
        desc [F,cap,32] u8, n [F] i32, depth [F,rows,cols] f32 -> dict(m [F,out_cap,2] f32, d [F,out_cap,32] u8,
        X, b [F,out_cap,3] f64, M [F] i32)."""
        _check(kp, torch.float32, "kp", (None, None, 2))
        F, cap = kp.shape[0], kp.shape[1]
        _check(desc, torch.uint8, "desc", (F, cap, 32))
        _check(n, torch.int32, "n", (F,))
        _check(depth, torch.float32, "depth", (F, None, None))
        rows, cols = depth.shape[1], depth.shape[2]
        dev = kp.device
        if out is None:
            out = dict(m=torch.zeros((F, out_cap, 2), dtype=torch.float32, device=dev),
                       d=torch.zeros((F, out_cap, 32), dtype=torch.uint8, device=dev),
                       X=torch.zeros((F, out_cap, 3), dtype=torch.float64, device=dev),
                       b=torch.zeros((F, out_cap, 3), dtype=torch.float64, device=dev),
                       M=torch.zeros((F,), dtype=torch.int32, device=dev))
        _check(out["m"], torch.float32, "m", (F, out_cap, 2))
        _check(out["d"], torch.uint8, "d", (F, out_cap, 32))
        _check(out["X"], torch.float64, "X", (F, out_cap, 3))
        _check(out["b"], torch.float64, "b", (F, out_cap, 3))
        _check(out["M"], torch.int32, "M", (F,))
        self._call(self._lib.sosvo_rgbd_assemble, ctypes.cast(ctypes.pointer(cam), c_p), _ptr(kp), _ptr(desc), _ptr(n),
                   _ptr(depth), F, rows, cols, cap, int(out_cap), _ptr(out["m"]), _ptr(out["d"]), _ptr(out["X"]),
                   _ptr(out["b"]), _ptr(out["M"]))
        return out

    def f2f_assemble_central(self, frames, ref_frame, cur_frame, keys, order, corr_cap, pct_good_matches=1.0,
                             max_hdiff=-1.0, out=None):
        """TrackerRGBDSE3.track_frame steps 1-2, batched.  frames: dict from rgbd_assemble; ref_frame / cur_frame
        [NP] i32; keys [NP, frame_cap, 1] u32, order [NP, frame_cap] i32 -> dict(f, p [NP,corr_cap,3] f64,
        q, t [NP,corr_cap] i32, n [NP] i32)."""
        NP = ref_frame.shape[0]
        frame_cap = frames["m"].shape[1]
        _check(ref_frame, torch.int32, "ref_frame", (NP,))
        _check(cur_frame, torch.int32, "cur_frame", (NP,))
        _check(keys, torch.uint32, "keys", (NP, frame_cap, 1))
        _check(order, torch.int32, "order", (NP, frame_cap))
        dev = ref_frame.device
        if out is None:
            out = dict(f=torch.zeros((NP, corr_cap, 3), dtype=torch.float64, device=dev),
                       p=torch.zeros((NP, corr_cap, 3), dtype=torch.float64, device=dev),
                       q=torch.zeros((NP, corr_cap), dtype=torch.int32, device=dev),
                       t=torch.zeros((NP, corr_cap), dtype=torch.int32, device=dev),
                       n=torch.zeros((NP,), dtype=torch.int32, device=dev))
        _check(out["f"], torch.float64, "f", (NP, corr_cap, 3))
        _check(out["p"], torch.float64, "p", (NP, corr_cap, 3))
        _check(out["q"], torch.int32, "q", (NP, corr_cap))
        _check(out["t"], torch.int32, "t", (NP, corr_cap))
        _check(out["n"], torch.int32, "n", (NP,))
        self._call(self._lib.sosvo_f2f_assemble_central, float(pct_good_matches), float(max_hdiff), _ptr(frames["m"]),
                   _ptr(frames["X"]), _ptr(frames["b"]), _ptr(frames["M"]), frame_cap, _ptr(ref_frame), _ptr(cur_frame),
                   _ptr(keys), _ptr(order), NP, int(corr_cap), _ptr(out["f"]), _ptr(out["p"]), _ptr(out["q"]),
                   _ptr(out["t"]), _ptr(out["n"]))
        return out

    # ---- whole hot path ------------------------------------------------------------------
    def frame_pair_batch_workspace(self, cfg):
        return int(self._lib.sosvo_frame_pair_batch_workspace(ctypes.cast(ctypes.pointer(cfg), c_p)))

    def frame_pair_batch_streams_workspace(self, cfg, n_streams):
        return int(self._lib.sosvo_frame_pair_batch_streams_workspace(ctypes.cast(ctypes.pointer(cfg), c_p), int(n_streams)))

    def frame_pair_batch_join(self):
        """sosvo_frame_pair_batch_streams_join: the context's stream waits for every part of the latest enqueue."""
        self._call(self._lib.sosvo_frame_pair_batch_streams_join)

    def frame_pair_batch(self, rig, cfg, omni, unwrap_table, mask_bits, pattern, workspace, results=None, n_streams=1,
                         join=True):
        """omni [2B,H,W,3] u8, unwrap_table [2,rows,cols,2] u32, mask_bits [2,rows,cols] u32, pattern [512,2] i8,
        workspace u8 [>= frame_pair_batch_workspace(cfg)] -> results [B,16] f64 (see include/sosvo.h).
        join=False (n_streams > 1 only): sosvo_frame_pair_batch_streams_enqueue -- call frame_pair_batch_join() before
        reading the records on this context's stream."""
        B = int(cfg.n_pairs)
        _check(omni, torch.uint8, "omni", (2 * B, cfg.H, cfg.W, 3))
        _check(unwrap_table, torch.uint32, "unwrap_table", (2, cfg.rows, cfg.cols, 2))
        _check(mask_bits, torch.uint32, "mask_bits", (2, cfg.rows, cfg.cols))
        _check(pattern, torch.int8, "pattern", (512, 2))
        _check(workspace, torch.uint8, "workspace", ndim=1)
        if results is None:
            results = torch.empty((B, 16), dtype=torch.float64, device=omni.device)
        _check(results, torch.float64, "results", (B, 16))
        if int(n_streams) > 1:   # the batch split over internal HIP streams of the library
            fn = self._lib.sosvo_frame_pair_batch_streams if join else self._lib.sosvo_frame_pair_batch_streams_enqueue
            self._call(fn, ctypes.cast(ctypes.pointer(rig), c_p),
                       ctypes.cast(ctypes.pointer(cfg), c_p), int(n_streams), _ptr(omni), _ptr(unwrap_table), _ptr(mask_bits),
                       _ptr(pattern), _ptr(workspace), int(workspace.numel()), _ptr(results))
            return results
        self._call(self._lib.sosvo_frame_pair_batch, ctypes.cast(ctypes.pointer(rig), c_p),
                   ctypes.cast(ctypes.pointer(cfg), c_p), _ptr(omni), _ptr(unwrap_table), _ptr(mask_bits), _ptr(pattern),
                   _ptr(workspace), int(workspace.numel()), _ptr(results))
        return results

    # ---- sequence mode (include/sosvo.h "Sequence mode") -----------------------------------------
    def sequence_workspace(self, cfg, window, slots):
        return int(self._lib.sosvo_sequence_workspace(ctypes.cast(ctypes.pointer(cfg), c_p), int(window), int(slots)))

    def sequence_front_end(self, rig, cfg, window, slots, omni, first_slot, unwrap_table, mask_bits, pattern, workspace):
        """omni [n,H,W,3] u8 (n <= window) -> store slots first_slot .. first_slot + n - 1 (asynchronous)."""
        _check(omni, torch.uint8, "omni", (None, cfg.H, cfg.W, 3))
        _check(unwrap_table, torch.uint32, "unwrap_table", (2, cfg.rows, cfg.cols, 2))
        _check(mask_bits, torch.uint32, "mask_bits", (2, cfg.rows, cfg.cols))
        _check(pattern, torch.int8, "pattern", (512, 2))
        _check(workspace, torch.uint8, "workspace", ndim=1)
        self._call(self._lib.sosvo_sequence_front_end, ctypes.cast(ctypes.pointer(rig), c_p), ctypes.cast(ctypes.pointer(cfg), c_p),
                   int(window), int(slots), _ptr(omni), int(omni.shape[0]), int(first_slot), _ptr(unwrap_table), _ptr(mask_bits),
                   _ptr(pattern), _ptr(workspace), int(workspace.numel()))

    def sequence_track(self, rig, cfg, window, slots, ref_slots, cur_slots, seed, workspace, results):
        """ref_slots / cur_slots: host int sequences of equal length n <= cfg.n_pairs -> results[:n] [n,16] f64 (device,
        asynchronous); pair i samples with seed + i."""
        n = len(ref_slots)
        if len(cur_slots) != n:
            raise SosvoError("ref_slots and cur_slots must have the same length")
        _check(workspace, torch.uint8, "workspace", ndim=1)
        _check(results, torch.float64, "results", (None, 16))
        if results.shape[0] < n:
            raise SosvoError("results holds fewer rows than slot pairs")
        ref = (ctypes.c_int32 * max(n, 1))(*[int(x) for x in ref_slots])
        cur = (ctypes.c_int32 * max(n, 1))(*[int(x) for x in cur_slots])
        self._call(self._lib.sosvo_sequence_track, ctypes.cast(ctypes.pointer(rig), c_p), ctypes.cast(ctypes.pointer(cfg), c_p),
                   int(window), int(slots), ctypes.cast(ref, c_p), ctypes.cast(cur, c_p), n, int(seed), _ptr(workspace),
                   int(workspace.numel()), _ptr(results))
        return results

    def sequence_copy_slot(self, cfg, window, slots, src, dst, workspace):
        self._call(self._lib.sosvo_sequence_copy_slot, ctypes.cast(ctypes.pointer(cfg), c_p), int(window), int(slots), int(src),
                   int(dst), _ptr(workspace), int(workspace.numel()))

    def sequence_frame_counts(self, cfg, window, slots, first_slot, n, workspace):
        """-> list of n ints (StereoPanoramicFrame.num_valid_keypoints of the slots); synchronises the stream."""
        out = (ctypes.c_int32 * max(int(n), 1))()
        self._call(self._lib.sosvo_sequence_frame_counts, ctypes.cast(ctypes.pointer(cfg), c_p), int(window), int(slots),
                   int(first_slot), int(n), _ptr(workspace), int(workspace.numel()), ctypes.cast(out, c_p))
        return [int(out[i]) for i in range(int(n))]

    # ---- RGB-D sequence mode ------------------------------------------------------------------------
    def rgbd_sequence_workspace(self, cfg, window, slots):
        return int(self._lib.sosvo_rgbd_sequence_workspace(ctypes.cast(ctypes.pointer(cfg), c_p), int(window), int(slots)))

    def rgbd_sequence_front_end(self, cam, cfg, window, slots, bgr, depth, first_slot, mask_bits, pattern, workspace):
        _check(bgr, torch.uint8, "bgr", (None, cfg.rows, cfg.cols, 3))
        _check(depth, torch.float32, "depth", (bgr.shape[0], cfg.rows, cfg.cols))
        _check(mask_bits, torch.uint32, "mask_bits", (1, cfg.rows, cfg.cols))
        _check(pattern, torch.int8, "pattern", (512, 2))
        _check(workspace, torch.uint8, "workspace", ndim=1)
        self._call(self._lib.sosvo_rgbd_sequence_front_end, ctypes.cast(ctypes.pointer(cam), c_p), ctypes.cast(ctypes.pointer(cfg), c_p),
                   int(window), int(slots), _ptr(bgr), _ptr(depth), int(bgr.shape[0]), int(first_slot), _ptr(mask_bits),
                   _ptr(pattern), _ptr(workspace), int(workspace.numel()))

    def rgbd_sequence_track(self, cfg, window, slots, ref_slots, cur_slots, seed, workspace, results):
        n = len(ref_slots)
        if len(cur_slots) != n:
            raise SosvoError("ref_slots and cur_slots must have the same length")
        _check(workspace, torch.uint8, "workspace", ndim=1)
        _check(results, torch.float64, "results", (None, 16))
        if results.shape[0] < n:
            raise SosvoError("results holds fewer rows than slot pairs")
        ref = (ctypes.c_int32 * max(n, 1))(*[int(x) for x in ref_slots])
        cur = (ctypes.c_int32 * max(n, 1))(*[int(x) for x in cur_slots])
        self._call(self._lib.sosvo_rgbd_sequence_track, ctypes.cast(ctypes.pointer(cfg), c_p), int(window), int(slots),
                   ctypes.cast(ref, c_p), ctypes.cast(cur, c_p), n, int(seed), _ptr(workspace), int(workspace.numel()), _ptr(results))
        return results

    def rgbd_sequence_copy_slot(self, cfg, window, slots, src, dst, workspace):
        self._call(self._lib.sosvo_rgbd_sequence_copy_slot, ctypes.cast(ctypes.pointer(cfg), c_p), int(window), int(slots), int(src),
                   int(dst), _ptr(workspace), int(workspace.numel()))

    def rgbd_sequence_frame_counts(self, cfg, window, slots, first_slot, n, workspace):
        out = (ctypes.c_int32 * max(int(n), 1))()
        self._call(self._lib.sosvo_rgbd_sequence_frame_counts, ctypes.cast(ctypes.pointer(cfg), c_p), int(window), int(slots),
                   int(first_slot), int(n), _ptr(workspace), int(workspace.numel()), ctypes.cast(out, c_p))
        return [int(out[i]) for i in range(int(n))]

    def rgbd_pair_batch_workspace(self, cfg):
        return int(self._lib.sosvo_rgbd_pair_batch_workspace(ctypes.cast(ctypes.pointer(cfg), c_p)))

    def rgbd_pair_batch(self, cam, cfg, bgr, depth, mask_bits, pattern, workspace, results=None):
        """bgr [2B,rows,cols,3] u8, depth [2B,rows,cols] f32, mask_bits [1,rows,cols] u32, pattern [512,2] i8,
        workspace u8 [>= rgbd_pair_batch_workspace(cfg)] -> results [B,16] f64 (see include/sosvo.h)."""
        B = int(cfg.n_pairs)
        _check(bgr, torch.uint8, "bgr", (2 * B, cfg.rows, cfg.cols, 3))
        _check(depth, torch.float32, "depth", (2 * B, cfg.rows, cfg.cols))
        _check(mask_bits, torch.uint32, "mask_bits", (1, cfg.rows, cfg.cols))
        _check(pattern, torch.int8, "pattern", (512, 2))
        _check(workspace, torch.uint8, "workspace", ndim=1)
        if results is None:
            results = torch.empty((B, 16), dtype=torch.float64, device=bgr.device)
        _check(results, torch.float64, "results", (B, 16))
        self._call(self._lib.sosvo_rgbd_pair_batch, ctypes.cast(ctypes.pointer(cam), c_p),
                   ctypes.cast(ctypes.pointer(cfg), c_p), _ptr(bgr), _ptr(depth), _ptr(mask_bits), _ptr(pattern),
                   _ptr(workspace), int(workspace.numel()), _ptr(results))
        return results

    def f2f_assemble(self, rig, frames, ref_frame, cur_frame, keys_top, order_top, keys_bot, order_bot, corr_cap,
                     out=None):
        """frames = dict from stereo_assemble; ref_frame / cur_frame [NP] i32; keys_* [NP, frame_cap, 1] u32,
        order_* [NP, frame_cap] i32 -> dict(f, p [NP,corr_cap,3] f64, cam, q, t [NP,corr_cap] i32, n, n_top [NP])."""
        F, frame_cap = frames["m_top"].shape[0], frames["m_top"].shape[1]
        NP = ref_frame.shape[0]
        _check(ref_frame, torch.int32, "ref_frame", (NP,))
        _check(cur_frame, torch.int32, "cur_frame", (NP,))
        _check(frames["m_top"], torch.float32, "m_top", (F, frame_cap, 2))
        _check(frames["m_bot"], torch.float32, "m_bot", (F, frame_cap, 2))
        for k in ("X", "b_top", "b_bot"):
            _check(frames[k], torch.float64, k, (F, frame_cap, 3))
        _check(frames["M"], torch.int32, "M", (F,))
        for name, t in (("keys_top", keys_top), ("keys_bot", keys_bot)):
            _check(t, torch.uint32, name, (NP, frame_cap, 1))
        for name, t in (("order_top", order_top), ("order_bot", order_bot)):
            _check(t, torch.int32, name, (NP, frame_cap))
        dev = ref_frame.device
        if out is None:
            out = dict(f=torch.zeros((NP, corr_cap, 3), dtype=torch.float64, device=dev),
                       p=torch.zeros((NP, corr_cap, 3), dtype=torch.float64, device=dev),
                       cam=torch.zeros((NP, corr_cap), dtype=torch.int32, device=dev),
                       q=torch.zeros((NP, corr_cap), dtype=torch.int32, device=dev),
                       t=torch.zeros((NP, corr_cap), dtype=torch.int32, device=dev),
                       n=torch.zeros((NP,), dtype=torch.int32, device=dev),
                       n_top=torch.zeros((NP,), dtype=torch.int32, device=dev))
        _check(out["f"], torch.float64, "f", (NP, corr_cap, 3))
        _check(out["p"], torch.float64, "p", (NP, corr_cap, 3))
        for k in ("cam", "q", "t"):
            _check(out[k], torch.int32, k, (NP, corr_cap))
        _check(out["n"], torch.int32, "n", (NP,))
        _check(out["n_top"], torch.int32, "n_top", (NP,))
        self._call(self._lib.sosvo_f2f_assemble, ctypes.cast(ctypes.pointer(rig), c_p), _ptr(frames["m_top"]),
                   _ptr(frames["m_bot"]), _ptr(frames["X"]), _ptr(frames["b_top"]), _ptr(frames["b_bot"]),
                   _ptr(frames["M"]), frame_cap, _ptr(ref_frame), _ptr(cur_frame), _ptr(keys_top), _ptr(order_top),
                   _ptr(keys_bot), _ptr(order_bot), NP, corr_cap, _ptr(out["f"]), _ptr(out["p"]), _ptr(out["cam"]),
                   _ptr(out["q"]), _ptr(out["t"]), _ptr(out["n"]), _ptr(out["n_top"]))
        return out
