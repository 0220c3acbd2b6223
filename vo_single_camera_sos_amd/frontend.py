"""Batched image front end: omni frames -> per-bucket keypoints + descriptors of both panoramas, i.e.
OmniStereoModel.set_current_omni_image (omnistereo/camera_models.py:3107-3120) followed by
OmniCamModel.detect_sparse_features_on_panorama for both mirrors (camera_models.py:1610-1797), for F frames
at once.  All buffers live in HBM; a run is five asynchronous C-ABI calls (K1, K2+K3, K4, K6)."""
import numpy as np
import torch

from . import orb_pattern
from .device import Context


class DeviceImageModel(object):
    """Per-model constants of the image stages, uploaded once: float32 unwrap maps and annulus masks of the two
    mirrors, the azimuthal masks as bit fields, the descriptor pattern."""

    def __init__(self, ctx, gums, omni_shape, azimuth_mask_degrees=30, overlap_degrees=0, elev_mask_padding=10,
                 stand_masks_azimuth_coord_in_degrees_list=(50, 170, 290), stand_masks_width_in_degrees=10,
                 mask_bits=None):
        """Defaults are TrackerStereoSE3.bootstrap_tracker's (pose_est_tools.py:870-878).  `mask_bits`
        [2, rows, cols] uint32 may be given to skip the mask construction."""
        assert isinstance(ctx, Context)
        top, bot = gums.top_model, gums.bot_model
        if top.mask is None or bot.mask is None:
            gums.make_annulus_masks(omni_shape)
        self.H, self.W = omni_shape
        self.rows, self.cols = top.panorama.rows, top.panorama.cols
        assert (bot.panorama.rows, bot.panorama.cols) == (self.rows, self.cols)
        maps = [m.panorama.float32_maps() for m in (top, bot)]
        dev = ctx.device
        self.map_x = torch.from_numpy(np.stack([maps[0][0], maps[1][0]])).to(dev)
        self.map_y = torch.from_numpy(np.stack([maps[0][1], maps[1][1]])).to(dev)
        self.omni_masks = torch.from_numpy(np.stack([top.mask, bot.mask])).to(dev)
        if mask_bits is None:
            bits = []
            for m in (top, bot):
                m.panorama.generate_azimuthal_masks(
                    azimuth_mask_degrees, overlap_degrees, elev_mask_padding=elev_mask_padding,
                    stand_masks_azimuth_coord_in_degrees_list=list(stand_masks_azimuth_coord_in_degrees_list),
                    stand_masks_width_in_degrees=stand_masks_width_in_degrees, omni_shape=omni_shape)
                bits.append(m.panorama.mask_bits())
            mask_bits = np.stack(bits)
            self.nmask = max(1, len(top.panorama.azimuthal_masks))
        else:
            self.nmask = int(np.log2(float(mask_bits.max()))) + 1 if mask_bits.max() > 0 else 1
        self.mask_bits_host = np.ascontiguousarray(mask_bits, dtype=np.uint32)
        self.mask_bits = torch.from_numpy(self.mask_bits_host).to(dev)
        self.pattern_host = orb_pattern.orb_pattern()
        self.pattern = torch.from_numpy(self.pattern_host).to(dev)


class ImageFrontEnd(object):
    def __init__(self, ctx, model, nframes, detection_method="GFT", num_of_features=1000, kp_cap=None,
                 median_win_size=11, quality=0.01, min_distance=5.0, edge=31, keep_panoramas=True,
                 skip_unreachable_rows=True):
        """num_of_features: per azimuthal mask (FeatureMatcher.num_of_features, pose_est_tools.py:862).
        keep_panoramas=False: K1 is fused into the median kernel and the colour panoramas are not materialised
        (nothing downstream of K3 reads them); identical gray images."""
        if detection_method.upper() not in ("GFT", "ORB", "FAST", "AGAST"):
            raise NotImplementedError("detection method %r: GFT (the reference default, pose_est_tools.py:684), ORB, "
                                      "FAST and AGAST are built" % detection_method)
        self.ctx, self.model, self.F = ctx, model, int(nframes)
        self.method = detection_method.upper()
        self.num_of_features, self.median_win_size = int(num_of_features), int(median_win_size)
        self.quality, self.min_distance, self.edge = float(quality), float(min_distance), int(edge)
        self.skip_unreachable_rows = bool(skip_unreachable_rows)
        if kp_cap:
            self.kp_cap = int(kp_cap)
        elif self.method in ("FAST", "AGAST"):  # every corner is kept (num_of_features only names the ORB descriptor object)
            self.kp_cap = 2048
        elif self.method == "ORB":  # retainBest keeps ties beyond the quota: leave head room
            self.kp_cap = int(min(2048, max(64, -(-int(self.num_of_features * 1.25) // 64) * 64)))
        else:
            self.kp_cap = int(min(1024, max(64, -(-self.num_of_features // 64) * 64)))
        dev, m = ctx.device, model
        NI, P = 2 * self.F, 2 * self.F * m.nmask
        self.omni = torch.zeros((self.F, m.H, m.W, 3), dtype=torch.uint8, device=dev)
        self.keep_panoramas = bool(keep_panoramas) or self.median_win_size not in (0, 1, 3, 5, 11)
        self.pano = torch.zeros((2, self.F, m.rows, m.cols, 3), dtype=torch.uint8, device=dev) if self.keep_panoramas else None
        self.gray = torch.zeros((NI, m.rows, m.cols), dtype=torch.uint8, device=dev)
        self.kp = torch.zeros((P, self.kp_cap, 2), dtype=torch.float32, device=dev)
        self.n = torch.zeros((P,), dtype=torch.int32, device=dev)
        self.status = torch.zeros((P,), dtype=torch.int32, device=dev)
        self.desc = torch.zeros((P, self.kp_cap, 32), dtype=torch.uint8, device=dev)
        self.cos_a, self.sin_a = orb_pattern.angle_cos_sin(orb_pattern.GFT_KEYPOINT_ANGLE)
        # Rows no consumer can reach (beyond the masks' elevation padding + the GFT halo and the descriptor border) are
        # not computed by the fused K1-K3 kernel.  Model constant, keyed by what it depends on; computed here (and
        # waited for) so that front ends on other streams that share the model find it ready.
        self.gray_rows = None
        if self.method == "GFT" and skip_unreachable_rows and not self.keep_panoramas:
            key = (self.edge, self.cos_a, self.sin_a)
            cache = m.__dict__.setdefault("_gray_rows", {})
            if key not in cache:
                cache[key] = ctx.gray_rows_needed(m.mask_bits, m.nmask, self.edge, m.pattern, self.cos_a, self.sin_a)
                ctx.synchronize()
            self.gray_rows = cache[key]
        if self.method == "ORB":
            self.kp4 = torch.zeros((P, self.kp_cap, 4), dtype=torch.float32, device=dev)
            self.resp = torch.zeros((P, self.kp_cap), dtype=torch.float32, device=dev)
            if getattr(model, "mask_pyr", None) is None:
                model.mask_pyr = ctx.orb_mask_pyramid(model.mask_bits, model.nmask)

    def load_frames(self, omni):
        """omni: numpy or torch uint8 [F, H, W, 3] (BGR) -> resident in HBM."""
        t = torch.from_numpy(np.ascontiguousarray(omni)) if isinstance(omni, np.ndarray) else omni
        self.omni.copy_(t.to(self.ctx.device))

    def run(self):
        self.run_images()
        self.run_features()

    def run_images(self):
        """K1 + K2 + K3: omni frames -> median-blurred gray panoramas (the VALU-bound half of the front end)."""
        c, m = self.ctx, self.model
        if getattr(m, "unwrap_table", None) is None:  # once per model
            m.unwrap_table = c.unwrap_prepare(m.omni_masks, m.map_x, m.map_y, (m.H, m.W))
        if self.keep_panoramas:
            c.unwrap_table(self.omni, m.unwrap_table, pano=self.pano)                                 # K1 (a1 + a2)
            c.median_gray(self.pano.view(2 * self.F, m.rows, m.cols, 3), self.median_win_size, gray=self.gray)  # K2 + K3
        else:
            c.unwrap_median_gray(self.omni, m.unwrap_table, self.median_win_size, gray=self.gray,
                                 row_range=self.gray_rows)                                           # K1 + K2 + K3

    def run_features(self):
        """K4 / K5 + K6: keypoints and descriptors per azimuthal mask on the gray panoramas."""
        c, m = self.ctx, self.model
        if self.method == "ORB":
            c.detect_describe_orb(self.gray, m.mask_pyr, self.F, m.nmask, self.num_of_features, m.pattern, self.kp4,
                                  self.resp, self.n, self.desc, kp_xy=self.kp)                   # K5 + K6' on one pyramid
            return
        if self.method in ("FAST", "AGAST"):
            (c.detect_fast if self.method == "FAST" else c.detect_agast)(
                self.gray, m.mask_bits, self.F, m.nmask, self.kp_cap, threshold=10, kp=self.kp, n=self.n,
                status=self.status)                                                               # FAST / AGAST + NMS
            c.describe_orb(self.gray, self.kp, self.n, m.nmask, m.pattern, self.cos_a, self.sin_a, edge=self.edge,
                           desc=self.desc)                                                        # K6
            return
        c.detect_gft(self.gray, m.mask_bits, self.F, m.nmask, self.kp_cap, quality=self.quality,
                     min_distance=self.min_distance, max_corners=self.num_of_features, kp=self.kp, n=self.n,
                     status=self.status)                                                          # K4
        c.describe_orb(self.gray, self.kp, self.n, m.nmask, m.pattern, self.cos_a, self.sin_a, edge=self.edge,
                       desc=self.desc, row_range=None if self.keep_panoramas else self.gray_rows)  # K6

    # views in the layout the matching stages expect: problem = frame * nmask + mask
    def view_arrays(self):
        h = self.F * self.model.nmask
        return dict(kp_top=self.kp[:h], kp_bot=self.kp[h:], desc_top=self.desc[:h], desc_bot=self.desc[h:],
                    n_top=self.n[:h], n_bot=self.n[h:])
