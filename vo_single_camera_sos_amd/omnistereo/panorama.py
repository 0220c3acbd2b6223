"""Host-side mirror of the hot-path part of the reference's `omnistereo.panorama.Panorama`: panorama
geometry, the unwrap look-up tables (built once per model with the same numpy float32/float64 operations
as the reference, so the tables are bit-identical), pixel -> direction angles, azimuthal bucket masks, and
`get_panoramic_image`, which runs on the GPU (libsosvo K1) -- there is no CPU remap here.

Reference: omnistereo/panorama.py:51-172 (dimensions), :414-484 (_generate_LUTs), :258-321
(get_panoramic_image), :616-666 (angles), :520-589 (generate_azimuthal_masks), :691-702."""
import numpy as np


class Panorama(object):
    def __init__(self, projection_model, **kwargs):
        self.model = projection_model
        self.name = projection_model.mirror_name + " panorama"
        self.omni_img = None
        self.panoramic_img = None
        self.cyl_radius = 1.0
        self.globally_highest_elevation_angle = projection_model.globally_highest_elevation_angle
        self.globally_lowest_elevation_angle = projection_model.globally_lowest_elevation_angle
        # panorama.py:142-146
        self.cyl_height_max = self.cyl_radius * np.tan(self.globally_highest_elevation_angle)
        self.z_height_min = self.cyl_radius * np.tan(self.globally_lowest_elevation_angle)
        self.cyl_height = self.cyl_height_max - self.z_height_min
        self.azimuthal_masks = []
        self.azimuthal_shift = kwargs.get("azimuthal_shift", 0)
        if "width" in kwargs:  # panorama.py:148-160: square pixels from the width
            self.width = kwargs["width"]
            self.cols = int(np.ceil(self.width))
            self.cyl_circumference = 2 * np.pi * self.cyl_radius
            self.pixel_size = self.cyl_circumference / float(self.cols)
            self.height = self.cyl_height / self.pixel_size
            self.rows = int(np.ceil(self.height))
        else:  # :161-170: from the height
            self.height = kwargs.get("height", 100)
            self.rows = int(np.ceil(self.height))
            self.cyl_circumference = 2 * np.pi * self.cyl_radius
            self.pixel_size = self.cyl_height / float(self.rows)
            self.width = self.cyl_circumference / self.pixel_size
            self.cols = int(np.ceil(self.width))
        self.aspect_ratio = float(self.cols) / float(self.rows)
        self._generate_LUTs()
        self._device_maps = None

    # ---- look-up tables (panorama.py:414-442,:478) ----
    def _generate_LUTs(self):
        m = self.model
        self.psi_LUT = np.linspace(0, 2 * np.pi, num=self.cols, endpoint=False)[::-1].copy()
        self.psi_LUT_2D = np.zeros((self.rows, self.cols), dtype="float32") + self.psi_LUT.astype("float32")
        cyl_height_LUT = np.linspace(self.cyl_height_max, self.z_height_min, num=self.rows, endpoint=False)
        self.theta_LUT = np.arctan2(cyl_height_LUT, self.cyl_radius)
        self.theta_LUT_validated = np.where(np.logical_and(m.lowest_elevation_angle <= self.theta_LUT,
                                                           self.theta_LUT <= m.highest_elevation_angle),
                                            self.theta_LUT, np.nan)
        self.theta_LUT_2D = np.zeros((self.rows, self.cols), dtype="float32") + \
            self.theta_LUT_validated[..., np.newaxis].astype("float32")
        with np.errstate(invalid="ignore"):
            self.world2cam_LUT_map_x, self.world2cam_LUT_map_y, _ = m.get_pixel_from_direction_angles(
                self.psi_LUT_2D, self.theta_LUT_2D)

    def float32_maps(self):
        """The maps as cv2.remap receives them (panorama.py:291-292)."""
        return self.world2cam_LUT_map_x.astype("float32"), self.world2cam_LUT_map_y.astype("float32")

    # ---- pixel -> angles (closed form, panorama.py:616-666) ----
    def get_elevation_from_panorama_row_without_LUT(self, row):
        if isinstance(row, np.ndarray):
            with np.errstate(invalid="ignore"):
                return np.where(np.logical_and(0. <= row, row < self.rows),
                                np.arctan2(self.cyl_height_max - self.pixel_size * row, self.cyl_radius), np.nan)
        if 0 <= row < self.rows:
            return np.arctan2(self.cyl_height_max - self.pixel_size * row, self.cyl_radius)
        return None

    def get_azimuth_from_panorama_col_without_LUT(self, col):
        if isinstance(col, np.ndarray):
            with np.errstate(invalid="ignore"):
                return np.where(np.logical_and(0. <= col, col < self.cols),
                                self.cyl_circumference - self.pixel_size * col, np.nan)
        if 0 <= col < self.cols:
            return self.cyl_circumference - self.pixel_size * col
        return None

    def get_direction_angles_from_pixel_pano(self, m_pano, use_LUTs=False):
        if use_LUTs:
            raise NotImplementedError("LUT-based angle look-up is not on the VO path (pose_est_tools.py:344-345)")
        return (self.get_azimuth_from_panorama_col_without_LUT(m_pano[..., 0]),
                self.get_elevation_from_panorama_row_without_LUT(m_pano[..., 1]))

    def get_panorama_col_from_azimuth(self, azimuth):
        """panorama.py:691-702"""
        arc_length = self.cyl_radius * np.mod(azimuth, 2.0 * np.pi)
        col_result = self.cols - 1 - int(np.uint(arc_length / self.pixel_size))
        return max(col_result, 0)

    # ---- unwrap (panorama.py:258-321) on the GPU ----
    def get_panoramic_image(self, input_omni_img, set_own=True, crop_out_bottom=False, border_RGB_color=None,
                            use_floating_point_prec=True):
        """input_omni_img: numpy [H,W,3] (or [H,W]) uint8 -> numpy panorama.  Runs libsosvo's K1 for this
        mirror only (the batched path unwraps both mirrors of many frames in one launch)."""
        if border_RGB_color not in (None, (0, 0, 0)):
            raise NotImplementedError("only the reference's default black border is supported")
        import torch
        from ..runtime import default_context
        ctx = default_context()
        img = np.ascontiguousarray(input_omni_img)
        gray_in = img.ndim == 2
        if gray_in:
            img = np.repeat(img[..., None], 3, axis=2)
        mx, my = self.float32_maps()
        dev = ctx.device
        t_maps = [torch.from_numpy(np.stack([a, a])).to(dev) for a in (mx, my)]
        pano = ctx.unwrap(torch.from_numpy(img[None]).to(dev), None, t_maps[0], t_maps[1])
        ctx.synchronize()
        out = pano[0, 0].cpu().numpy()
        if gray_in:
            out = np.ascontiguousarray(out[..., 0])
        if set_own:
            self.omni_img = input_omni_img
            self.panoramic_img = out
        return out

    def set_panoramic_image(self, omni_img, idx=-1, view=False, win_name_modifier="", border_RGB_color=None):
        return self.get_panoramic_image(omni_img, set_own=True, border_RGB_color=border_RGB_color)

    # ---- azimuthal bucket masks (panorama.py:520-589) ----
    def generate_azimuthal_masks(self, azimuth_mask_degrees, overlap_degrees=0, mask_also_on_elev=True,
                                 elev_mask_padding=0, stand_masks_azimuth_coord_in_degrees_list=(),
                                 stand_masks_width_in_degrees=1, show=False, omni_shape=None, unwrap_fn=None):
        """`unwrap_fn(omni_mask) -> pano_mask` overrides the GPU unwrap of the elevation mask (tests use it to
        build the masks without a GPU); default is self.get_panoramic_image."""
        # the same request as last time (every tracker bootstrap asks again, pose_est_tools.py:870-878): the masks are a pure
        # function of these arguments and of the model's annulus mask -- keep them (a GPU unwrap + twelve full-size images)
        mm = getattr(self.model, "mask", None)
        key = (float(azimuth_mask_degrees), float(overlap_degrees), bool(mask_also_on_elev), int(elev_mask_padding),
               tuple(float(a) for a in stand_masks_azimuth_coord_in_degrees_list), float(stand_masks_width_in_degrees),
               None if omni_shape is None else tuple(omni_shape), unwrap_fn is None, self.rows, self.cols,
               None if mm is None else (mm.shape, int(np.count_nonzero(mm))),
               tuple(float(v) for v in (self.model.inner_img_radius, self.model.outer_img_radius)))
        if getattr(self, "_azimuthal_masks_key", None) == key and self.azimuthal_masks:
            return self.azimuthal_masks
        self._azimuthal_masks_key = None
        azimuth_mask_radians = np.deg2rad(azimuth_mask_degrees)
        overlap_radians = np.deg2rad(overlap_degrees)
        self.azimuthal_masks = []
        pano_img_mask = None
        if mask_also_on_elev:
            if omni_shape is None:
                omni_shape = (self.model.image_size[1], self.model.image_size[0])
            if elev_mask_padding > 0:
                omni_img_mask = self.model.make_mask(mask_shape=omni_shape, radius_pixel_shrinking=elev_mask_padding)
            else:
                omni_img_mask = self.model.mask if self.model.mask is not None else self.model.make_mask(omni_shape)
            if unwrap_fn is not None:
                pano_img_mask = unwrap_fn(omni_img_mask)
            else:
                pano_img_mask = self.get_panoramic_image(omni_img_mask, set_own=False)

        def paint(mask, c1, c2, value):  # cv2.rectangle(..., thickness=-1): both corner columns included
            lo, hi = (c1, c2) if c1 <= c2 else (c2, c1)
            mask[:, lo:hi + 1] = value

        mask_for_stands = None
        if len(stand_masks_azimuth_coord_in_degrees_list) > 0:
            mask_for_stands = np.zeros((self.rows, self.cols), dtype=np.uint8) + 255
            half = np.deg2rad(stand_masks_width_in_degrees / 2.0)
            for azim_coord in np.deg2rad(stand_masks_azimuth_coord_in_degrees_list):
                a0, a1 = azim_coord - half, azim_coord + half
                c1 = self.get_panorama_col_from_azimuth(0 if a0 <= 0 else a0)
                c2 = 0 if a1 >= 2.0 * np.pi else self.get_panorama_col_from_azimuth(a1)
                paint(mask_for_stands, c1, c2, 0)
        for d in np.arange(start=0, stop=2 * np.pi - azimuth_mask_radians / 2., step=azimuth_mask_radians):
            a0, a1 = d - overlap_radians, d + azimuth_mask_radians + overlap_radians
            c1 = self.get_panorama_col_from_azimuth(0 if a0 <= 0 else a0)
            c2 = 0 if a1 >= 2.0 * np.pi else self.get_panorama_col_from_azimuth(a1)
            mask = np.zeros((self.rows, self.cols), dtype=np.uint8)
            paint(mask, c1, c2, 255)
            if pano_img_mask is not None:
                mask = np.where(mask > 0, mask & pano_img_mask, 0).astype(np.uint8)
            if mask_for_stands is not None:
                mask = np.where(mask > 0, mask & mask_for_stands, 0).astype(np.uint8)
            self.azimuthal_masks.append(mask)
        self._azimuthal_masks_key = key
        return self.azimuthal_masks

    def mask_bits(self):
        """The azimuthal masks as one uint32 bit field per pixel (bit m set = pixel belongs to mask m),
        the form the detection kernels consume.  No masks -> every pixel belongs to mask 0."""
        if not self.azimuthal_masks:
            return np.ones((self.rows, self.cols), dtype=np.uint32)
        if len(self.azimuthal_masks) > 32:
            raise ValueError("at most 32 azimuthal masks are supported")
        bits = np.zeros((self.rows, self.cols), dtype=np.uint32)
        for m, mask in enumerate(self.azimuthal_masks):
            bits |= (mask != 0).astype(np.uint32) << np.uint32(m)
        return bits
