"""Host-side mirror of the hot-path part of the reference's `omnistereo.gum`: the Generalized Unified
Model of ONE mirror (forward projection that defines the unwrap LUT, angles -> bearing) and the
two-mirror `GUMStereo` rig.  Same class / method names and argument meaning as the reference for the
methods the VO path uses; calibration, Jacobians and the pixel-lifting used only at calibration time are
out of scope.  numpy only: these run once per model, not per frame.

Reference: omnistereo/gum.py:2512-2562 (get_pixel_from_3D_point_wrt_M), :1368-1385, :2942-2971,
:2564-2575; omnistereo/camera_models.py:1031-1065 (map_angles_to_unit_sphere), :203-212."""
import numpy as np


def get_normalized_points(points_wrt_M):
    """camera_models.py:203-212"""
    pts = np.asarray(points_wrt_M)
    return pts[..., :3] / (np.linalg.norm(pts[..., :3], axis=-1)[..., np.newaxis])


class GUMParams(object):
    """The calibrated numbers of one mirror (reference: gum.Parameters fields, gum.py:77-116,:169-214)."""

    def __init__(self, xi1=0.0, xi2=0.0, xi3=1.0, k1=0.0, k2=0.0, k3=0.0, gamma1=1.0, gamma2=1.0, alpha_c=0.0,
                 u_center=0.0, v_center=0.0, use_distortion=True):
        self.xi1, self.xi2, self.xi3 = float(xi1), float(xi2), float(xi3)
        self.k1, self.k2, self.k3 = float(k1), float(k2), float(k3)
        self.gamma1, self.gamma2, self.alpha_c = float(gamma1), float(gamma2), float(alpha_c)
        self.u_center, self.v_center = float(u_center), float(v_center)
        self.use_distortion = bool(use_distortion)
        self.p1 = self.p2 = 0.0                    # tangential terms: stored by the calibration toolbox, unused by the model
        self.roi_min_x = self.roi_min_y = self.roi_max_x = self.roi_max_y = None

    # The calibration toolbox's pre-calibration .bin (gum.py:216-272): native-endian float64 values, in order
    #   xi1 xi2 xi3 (new method) | xi3 (old)  k1 k2 p1 p2 k3  gamma1 gamma2 u_center v_center alpha_c  roi_min_x roi_min_y roi_max_x roi_max_y
    # with the centre and the ROI counted from 1 (MATLAB) -- the reader subtracts 1.
    _BIN_TAIL = ("k1", "k2", "p1", "p2", "k3", "gamma1", "gamma2", "u_center", "v_center", "alpha_c",
                 "roi_min_x", "roi_min_y", "roi_max_x", "roi_max_y")
    _BIN_ONE_BASED = ("u_center", "v_center", "roi_min_x", "roi_min_y", "roi_max_x", "roi_max_y")

    @classmethod
    def from_precalibration_bin(cls, filename, new_method=True, z_axis=1.0):
        """Reads a pre-calibration file.  A file that is too short raises ValueError (the reference prints the
        error and carries on with half-initialised parameters, gum.py:273-275)."""
        import struct
        n_head = 3 if new_method else 1
        n = n_head + len(cls._BIN_TAIL)
        with open(filename, "rb") as f:
            block = f.read(8 * n)
        if len(block) < 8 * n:
            raise ValueError("%s: %d bytes, a %s-method pre-calibration file has %d" % (filename, len(block),
                                                                                        "new" if new_method else "old", 8 * n))
        vals = struct.unpack("%dd" % n, block)
        out = cls(xi3=1.0 * z_axis)
        if new_method:
            out.xi1, out.xi2, out.xi3 = vals[:3]
        else:
            out.xi1, out.xi2, out.xi3 = 0.0, 0.0, vals[0]
        for k, v in zip(cls._BIN_TAIL, vals[n_head:]):
            setattr(out, k, v - 1.0 if k in cls._BIN_ONE_BASED else v)
        return out

    def to_precalibration_bin(self, filename, new_method=True):
        """The inverse of from_precalibration_bin (same layout, centre / ROI back to 1-based)."""
        import struct
        head = [self.xi1, self.xi2, self.xi3] if new_method else [self.xi3]
        tail = []
        for k in self._BIN_TAIL:
            v = getattr(self, k)
            v = 0.0 if v is None else float(v)
            tail.append(v + 1.0 if k in self._BIN_ONE_BASED else v)
        with open(filename, "wb") as f:
            f.write(struct.pack("%dd" % (len(head) + len(tail)), *(head + tail)))


class GUM(object):
    """One mirror.  `F` is the focus (viewpoint) wrt the common frame [C]; `z_axis` is +1 for the top
    mirror, -1 for the bottom one (gum.py:335-341)."""

    def __init__(self, params, z_axis, F, lowest_elevation_angle, highest_elevation_angle, inner_img_radius,
                 outer_img_radius, center_point=None, image_size=(640, 480), mirror_name="", units="mm",
                 center_point_inner=None, center_point_outer=None):
        """center_point: centre the radial limits were measured from (default: the principal point);
        center_point_inner / center_point_outer: centres of the inner and the outer circle of the mirror's annulus mask
        (camera_models.py:972-988, :1556-1557: a calibrated rig has one pair per mirror; default: center_point)."""
        self.precalib_params = params
        self.z_axis = float(z_axis)
        self.mirror_name = mirror_name or ("top" if z_axis > 0 else "bottom")
        self.units = units
        self.image_size = tuple(image_size)
        self.F = np.array([[F[0]], [F[1]], [F[2]], [1.0]], dtype=np.float64)
        self.Cp_wrt_M = [params.xi1, params.xi2, params.xi3]
        self.lowest_elevation_angle = float(lowest_elevation_angle)
        self.highest_elevation_angle = float(highest_elevation_angle)
        self.globally_lowest_elevation_angle = self.lowest_elevation_angle
        self.globally_highest_elevation_angle = self.highest_elevation_angle
        self.inner_img_radius, self.outer_img_radius = inner_img_radius, outer_img_radius
        c = (params.u_center, params.v_center) if center_point is None else center_point
        self.center_point = np.asarray(c, dtype=np.float64)
        self.center_point_inner = np.asarray(self.center_point if center_point_inner is None else center_point_inner, dtype=np.float64)
        self.center_point_outer = np.asarray(self.center_point if center_point_outer is None else center_point_outer, dtype=np.float64)
        # (the reference keeps them on the parameter object, camera_models.py:972-988; plain tuples here)
        params.center_point_inner = tuple(float(v) for v in self.center_point_inner)
        params.center_point_outer = tuple(float(v) for v in self.center_point_outer)
        self.T_model_wrt_C = np.identity(4)
        self.T_C_wrt_model = np.identity(4)
        self.set_pose(self.F[:3, 0], np.identity(3))
        self.mask = None
        self.panorama = None
        self.current_omni_img = None

    def set_pose(self, translation, rotation_matrix):
        """camera_models.py:955-962: pose of the model frame [M] wrt the common frame [C]."""
        self.T_model_wrt_C = np.identity(4)
        self.T_model_wrt_C[:3, :3] = rotation_matrix
        self.T_model_wrt_C[:3, 3] = translation
        self.T_C_wrt_model = np.linalg.inv(self.T_model_wrt_C)

    # ---- angles -> sphere (camera_models.py:1031-1065) ----
    def map_angles_to_unit_sphere(self, theta, psi):
        if isinstance(theta, np.ndarray):
            valid = np.logical_not(np.isnan(theta))
            b = np.where(valid, np.cos(theta), np.nan)
            z = np.where(valid, np.sin(theta), np.nan)
        else:
            b, z = np.cos(theta), np.sin(theta)
        x = b * np.cos(psi)
        y = b * np.sin(psi)
        return np.dstack((x, y, z, np.ones_like(x)))

    def get_3D_point_from_angles_wrt_focus(self, azimuth, elevation):
        """gum.py:2564-2575"""
        return self.map_angles_to_unit_sphere(elevation, azimuth)

    # ---- forward projection (gum.py:2512-2562) ----
    def get_pixel_from_3D_point_wrt_M(self, Pw_wrt_M, visualize=False):
        p = self.precalib_params
        Ps = get_normalized_points(Pw_wrt_M)                      # onto the unit sphere
        Ps_wrt_Cp = Ps - np.array(self.Cp_wrt_M)                  # gum.py:1368-1371
        p_und = (Ps_wrt_Cp / np.abs(Ps_wrt_Cp[..., 2][..., np.newaxis]))[..., :2]  # :1378-1381
        if p.use_distortion:                                       # :2942-2971
            rho_sq = p_und[..., 0] ** 2 + p_und[..., 1] ** 2
            factor = np.ones_like(rho_sq)
            for idx, k in enumerate((p.k1, p.k2, p.k3)):
                factor += k * rho_sq ** (idx + 1)
            p_dist = p_und * factor[..., np.newaxis]
        else:
            p_dist = p_und
        x, y = p_dist[..., 0], p_dist[..., 1]
        u = p.gamma1 * x + p.gamma1 * p.alpha_c * y + p.u_center   # :2554-2562
        v = p.gamma2 * y + p.v_center
        return u, v, np.dstack((u, v, np.ones_like(u)))

    def get_points_wrt_M(self, points_wrt_C_homo):
        return np.einsum("ij, klj->kli", self.T_C_wrt_model, points_wrt_C_homo)

    def get_pixel_from_3D_point_wrt_C(self, Pw_wrt_C, visualize=False):
        return self.get_pixel_from_3D_point_wrt_M(self.get_points_wrt_M(Pw_wrt_C))

    def get_pixel_from_direction_angles(self, azimuth, elevation, visualize=False):
        """camera_models.py:1067-1078"""
        return self.get_pixel_from_3D_point_wrt_M(self.get_3D_point_from_angles_wrt_focus(azimuth=azimuth,
                                                                                          elevation=elevation))

    # ---- masks (camera_models.py:1546-1569; discs are x^2 + y^2 <= r^2) ----
    # ---- per-frame API (camera_models.py:1610-1797, :1544-1580) on the GPU ----
    def set_omni_image(self, img, pano_width_in_pixels=1200, generate_panorama=False, idx=-1, view=False, apply_mask=True,
                       mask_RGB=None):
        self.current_omni_img = img
        if self.panorama is None:
            from .panorama import Panorama
            self.panorama = Panorama(self, width=pano_width_in_pixels)
        if generate_panorama:
            self.panorama.set_panoramic_image(img)

    def detect_sparse_features_on_panorama(self, feature_detection_method="ORB", num_of_features=50, median_win_size=0,
                                           show=True):
        """-> (list of keypoint lists, list of descriptor arrays), one entry per azimuthal mask.  The image stages
        of both mirrors run in one batched pass of the rig's device front end (results cached per omni image)."""
        rig = getattr(self, "rig", None)
        if rig is None:
            raise RuntimeError("detect_sparse_features_on_panorama: the model must belong to a GUMStereo rig")
        kp_lists, desc_lists = rig._detect_both(feature_detection_method, num_of_features, median_win_size)
        return kp_lists[self.view_index], desc_lists[self.view_index]

    @staticmethod
    def _disc(shape, center, radius):
        yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
        return (xx - int(center[0])) ** 2 + (yy - int(center[1])) ** 2 <= int(radius) ** 2

    def make_mask(self, mask_shape, radius_pixel_shrinking=0):
        r_in = self.inner_img_radius + radius_pixel_shrinking
        r_out = self.outer_img_radius - radius_pixel_shrinking
        m = self._disc(mask_shape, self.center_point_outer, r_out)     # camera_models.py:1556-1567
        if r_in > 0:
            m &= ~self._disc(mask_shape, self.center_point_inner, r_in)
        return m.astype(np.uint8) * 255


class GUMStereo(object):
    """The two-mirror rig (reference: gum.GUMStereo / camera_models.OmniStereoModel)."""

    def __init__(self, top_model, bottom_model, units="mm"):
        self.top_model, self.bot_model = top_model, bottom_model
        self.units = units
        hi = max(top_model.highest_elevation_angle, bottom_model.highest_elevation_angle)  # camera_models.py:2795-2800
        lo = min(top_model.lowest_elevation_angle, bottom_model.lowest_elevation_angle)
        for m in (top_model, bottom_model):
            m.globally_highest_elevation_angle, m.globally_lowest_elevation_angle = hi, lo
        self.baseline = self.get_baseline()
        self.current_omni_img = None
        self.construct_new_mask = True
        self.feature_matcher_for_static_stereo = None
        self.feature_matcher_for_motion = None
        self.T_Cest_wrt_Rgt = None  # hand-eye transformation to a ground-truth rig frame (camera_models.py:2826)
        top_model.rig, top_model.view_index = self, 0
        bottom_model.rig, bottom_model.view_index = self, 1
        # pose of the bottom model wrt the top one (camera_models.py:2811-2813)
        self.T_bot_wrt_top = top_model.T_C_wrt_model.dot(bottom_model.T_model_wrt_C)
        self._dev_model = None
        self._front_ends = {}
        self._omni_serial = 0
        self._detect_cache = (None, None)

    # ---- per-frame API on the GPU -------------------------------------------------------------------------
    def _context(self):
        from ..runtime import default_context
        return default_context()

    def _device_model(self):
        """Model constants in HBM (unwrap maps, annulus masks, azimuthal bucket masks): built once."""
        if self._dev_model is None:
            from ..frontend import DeviceImageModel
            shape = self.current_omni_img.shape[:2]
            bits = None
            if all(len(m.panorama.azimuthal_masks) > 0 for m in (self.top_model, self.bot_model)):
                bits = np.stack([m.panorama.mask_bits() for m in (self.top_model, self.bot_model)])
            elif not any(len(m.panorama.azimuthal_masks) > 0 for m in (self.top_model, self.bot_model)):
                rows, cols = self.top_model.panorama.rows, self.top_model.panorama.cols
                bits = np.ones((2, rows, cols), dtype=np.uint32)  # no buckets: one mask covering the panorama
            self._dev_model = DeviceImageModel(self._context(), self, shape, mask_bits=bits)
        return self._dev_model

    def _front_end(self, method, num_of_features, median_win_size):
        key = (str(method).upper(), int(num_of_features), int(median_win_size))
        if key not in self._front_ends:
            from ..frontend import ImageFrontEnd
            cap = int(min(4096, max(64, -(-int(num_of_features * (1.25 if key[0] == "ORB" else 1.0)) // 64) * 64)))
            self._front_ends[key] = ImageFrontEnd(self._context(), self._device_model(), 1, detection_method=key[0],
                                                  num_of_features=key[1], kp_cap=cap, median_win_size=key[2])
        return self._front_ends[key]

    def set_current_omni_image(self, img, pano_width_in_pixels=1200, generate_panoramas=False, idx=-1, view=False,
                               apply_mask=True, mask_RGB=None):
        """camera_models.py:3107-3120.  The annulus masks are part of the unwrap table (K1), so the masked omni
        images are never materialised; with generate_panoramas the two panoramas come back to the host."""
        self.current_omni_img = img
        self._omni_serial += 1
        for m in (self.top_model, self.bot_model):
            m.set_omni_image(img, pano_width_in_pixels=pano_width_in_pixels, generate_panorama=False)
        if generate_panoramas:
            import torch
            ctx, dm = self._context(), self._device_model()
            if getattr(dm, "unwrap_table", None) is None:
                dm.unwrap_table = ctx.unwrap_prepare(dm.omni_masks if apply_mask else None, dm.map_x, dm.map_y, (dm.H, dm.W))
            omni = torch.from_numpy(np.ascontiguousarray(img)[None]).to(ctx.device)
            pano = ctx.unwrap_table(omni, dm.unwrap_table)
            ctx.synchronize()
            host = pano.cpu().numpy()
            self.top_model.panorama.panoramic_img = host[0, 0]
            self.bot_model.panorama.panoramic_img = host[1, 0]
            self.top_model.panorama.omni_img = self.bot_model.panorama.omni_img = img

    def _detect_both(self, method, num_of_features, median_win_size):
        token = (self._omni_serial, str(method).upper(), int(num_of_features), int(median_win_size))
        if self._detect_cache[0] == token:
            return self._detect_cache[1]
        from .camera_models import KeyPoint
        fe = self._front_end(method, num_of_features, median_win_size)
        fe.load_frames(np.ascontiguousarray(self.current_omni_img)[None])
        fe.run()
        fe.ctx.synchronize()
        nm = fe.model.nmask
        kp, n, desc = fe.kp.cpu().numpy(), fe.n.cpu().numpy(), fe.desc.cpu().numpy()
        kp_lists, desc_lists = [[], []], [[], []]
        for view in range(2):
            for m in range(nm):
                p = view * nm + m
                cnt = int(n[p])
                kp_lists[view].append([KeyPoint(x, y, size=1.0) for x, y in kp[p, :cnt]])
                desc_lists[view].append(np.ascontiguousarray(desc[p, :cnt]))
        self._detect_cache = (token, (kp_lists, desc_lists))
        return self._detect_cache[1]

    def match_features_panoramic_top_bottom(self, keypts_list_top, desc_list_top, keypts_list_bot, desc_list_bot,
                                            min_rectified_disparity=1, max_horizontal_diff=1, show_matches=False,
                                            win_name="Matches"):
        """camera_models.py:3027-3101: per bucket, match bottom (query) against top (train), keep the first
        percentage_good_matches of the sorted matches, concatenate, gate the pixel pairs."""
        from .camera_models import keypoints_to_array
        from .common_cv import filter_pixel_correspondences
        fm = self.feature_matcher_for_static_stereo
        sel_top, sel_bot, dsel_top, dsel_bot = [], [], [], []
        buckets = [b for b in zip(keypts_list_top, desc_list_top, keypts_list_bot, desc_list_bot) if len(b[0]) and len(b[2])]
        matched = fm.match_arrays_many([(bot_d, top_d) for _, top_d, _, bot_d in buckets])   # all buckets in one launch
        for (top_k, top_d, bot_k, bot_d), (q, t, _) in zip(buckets, matched):
            good = int(fm.percentage_good_matches * len(q))
            if good > 0:
                sel_top.append(np.array(top_k, dtype=object)[t[:good]])
                sel_bot.append(np.array(bot_k, dtype=object)[q[:good]])
                dsel_top.append(np.asarray(top_d)[t[:good]])
                dsel_bot.append(np.asarray(bot_d)[q[:good]])
        if not sel_top:
            e = np.empty((0,), dtype=object)
            return (np.empty((0, 3)), e, np.empty((0, 32), np.uint8)), (np.empty((0, 3)), e, np.empty((0, 32), np.uint8)), \
                np.empty((0, 3), np.uint8)
        k_top, k_bot = np.concatenate(sel_top), np.concatenate(sel_bot)
        d_top, d_bot = np.concatenate(dsel_top), np.concatenate(dsel_bot)
        p_top = keypoints_to_array(k_top).astype(np.float64)
        p_bot = keypoints_to_array(k_bot).astype(np.float64)
        ok = filter_pixel_correspondences(matched_points_top=p_top, matched_points_bot=p_bot,
                                          min_rectified_disparity=min_rectified_disparity,
                                          max_horizontal_diff=max_horizontal_diff)
        colors = np.random.randint(low=0, high=256, size=(int(ok.sum()), 3), dtype="uint8")
        m_top = np.hstack((p_top[ok], np.ones((int(ok.sum()), 1))))
        m_bot = np.hstack((p_bot[ok], np.ones((int(ok.sum()), 1))))
        return (m_top, k_top[ok], d_top[ok]), (m_bot, k_bot[ok], d_bot[ok]), colors

    def get_triangulated_point_from_direction_angles(self, dir_angs_top, dir_angs_bot, use_midpoint_triangulation=True):
        """camera_models.py:3323-3364 -> [1, m, 4] homogeneous points wrt [C] (midpoint of the common
        perpendicular of the two rays; the only method the VO path uses)."""
        import torch
        if not use_midpoint_triangulation:
            raise NotImplementedError("only the midpoint triangulation (the VO default) is built")
        ctx = self._context()
        az1, el1 = [np.asarray(a, dtype=np.float64).reshape(-1) for a in dir_angs_top]
        az2, el2 = [np.asarray(a, dtype=np.float64).reshape(-1) for a in dir_angs_bot]
        out = np.ones((1, az1.shape[0], 4))
        if az1.shape[0]:
            t = [torch.from_numpy(np.ascontiguousarray(a)).to(ctx.device) for a in (az1, el1, az2, el2)]
            X = ctx.triangulate_midpoint(t[0], t[1], t[2], t[3], self.top_model.F[:3, 0], self.bot_model.F[:3, 0])
            ctx.synchronize()
            out[0, :, :3] = X.cpu().numpy()
        return out

    def filter_panoramic_points_due_to_range(self, xyz_points_wrt_C, min_3D_range=0, max_3D_range=0):
        """camera_models.py:3299-3321 -> bool [m].  As the reference, the norm runs over ALL columns of the rows it
        is given: the VO path passes homogeneous rows (pose_est_tools.py:372), so the trailing 1 is included."""
        import torch
        ctx = self._context()
        pts = np.asarray(xyz_points_wrt_C, dtype=np.float64)
        if pts.shape[-1] != 4:
            raise NotImplementedError("the GPU range filter takes the homogeneous [m,4] rows the VO path passes")
        pts = pts.reshape(-1, 4)
        if pts.shape[0] == 0:
            return np.zeros((0,), dtype=bool)
        ok = ctx.range_filter(torch.from_numpy(np.ascontiguousarray(pts[:, :3])).to(ctx.device), min_3D_range, max_3D_range)
        ctx.synchronize()
        return ok.cpu().numpy().astype(bool)

    def get_baseline(self):
        return self.top_model.F[2, 0] - self.bot_model.F[2, 0]

    def make_annulus_masks(self, shape):
        """camera_models.py:2944-2988: top = outer disc minus inner disc minus the bottom mirror's outer disc;
        bottom = its outer disc intersected with the top mirror's inner disc, minus its own inner disc."""
        t, b = self.top_model, self.bot_model
        # circle centres exactly as the reference picks them (:2946-2957, :2966-2983): each mirror's own inner / outer centre
        top = GUM._disc(shape, t.center_point_outer, t.outer_img_radius)
        if t.inner_img_radius > 0:
            top &= ~GUM._disc(shape, t.center_point_inner, t.inner_img_radius)
            if b.outer_img_radius > 0:
                top &= ~GUM._disc(shape, b.center_point_outer, b.outer_img_radius)
        bot = GUM._disc(shape, b.center_point_outer, b.outer_img_radius) & GUM._disc(shape, b.center_point_inner, t.inner_img_radius)
        bot &= ~GUM._disc(shape, b.center_point_inner, b.inner_img_radius)
        t.mask, b.mask = top.astype(np.uint8) * 255, bot.astype(np.uint8) * 255
        self.construct_new_mask = False
        return t.mask, b.mask


def synthetic_gums(scale=1.0, units="mm"):
    """The synthetic rig validated against the reference's numpy code:
    640x480 * scale image, top xi3 = +0.9 gamma 150, bottom xi3 = -0.9 gamma 60, foci at z = 150 / 50 mm.
    Elevation limits are the reference's own values for this model (they do not depend on `scale`)."""
    c = (319.5 * scale + (scale - 1.0) * 0.5, 239.5 * scale + (scale - 1.0) * 0.5)
    size = (int(640 * scale), int(480 * scale))
    top = GUM(GUMParams(xi3=+0.9, gamma1=150.0 * scale, gamma2=150.0 * scale, u_center=c[0], v_center=c[1]), +1.0,
              (0.0, 0.0, 150.0), -0.35290654146694395, 0.2618998070797145, 113 * scale, 226 * scale, image_size=size,
              units=units)
    bot = GUM(GUMParams(xi3=-0.9, gamma1=60.0 * scale, gamma2=60.0 * scale, u_center=c[0], v_center=c[1]), -1.0,
              (0.0, 0.0, 50.0), -0.3487218912619687, 0.26202807633801434, 50 * scale, 101 * scale, image_size=size,
              units=units)
    return GUMStereo(top, bot, units=units)


# ---- calibrated-model files --------------------------------------------------------------------------------
# The reference stores a calibrated rig as a pickle of live Python objects (common_tools.py:131,
# demo_vo_sos.py:119).  Here a rig is a small JSON document of the calibrated numbers:
# per mirror the GUM parameters (gum.py:77-116), focus, elevation limits, image radii and centre; plus units
# and the panorama width.
_PARAM_FIELDS = ("xi1", "xi2", "xi3", "k1", "k2", "k3", "gamma1", "gamma2", "alpha_c", "u_center", "v_center",
                 "use_distortion")


def gums_to_dict(gums):
    def mirror(m):
        d = {k: getattr(m.precalib_params, k) for k in _PARAM_FIELDS}
        if not np.array_equal(m.T_model_wrt_C[:3, :3], np.identity(3)):
            d["R_model_wrt_C"] = [[float(v) for v in row] for row in m.T_model_wrt_C[:3, :3]]
        d.update(z_axis=m.z_axis, F=[float(v) for v in m.F[:3, 0]], lowest_elevation_angle=m.lowest_elevation_angle,
                 highest_elevation_angle=m.highest_elevation_angle, inner_img_radius=float(m.inner_img_radius),
                 outer_img_radius=float(m.outer_img_radius), center_point=[float(v) for v in m.center_point],
                 center_point_inner=[float(v) for v in m.center_point_inner],
                 center_point_outer=[float(v) for v in m.center_point_outer], image_size=[int(v) for v in m.image_size])
        return d
    width = gums.top_model.panorama.cols if gums.top_model.panorama is not None else 1200
    return dict(format="sosvo-gums-1", units=gums.units, panorama_width=int(width), top=mirror(gums.top_model),
                bottom=mirror(gums.bot_model))


def gums_from_dict(d, with_panoramas=True):
    from .panorama import Panorama
    if d.get("format") != "sosvo-gums-1":
        raise ValueError("not a sosvo GUMS description (format %r)" % d.get("format"))

    def mirror(m):
        params = GUMParams(**{k: m[k] for k in _PARAM_FIELDS})
        g = GUM(params, m["z_axis"], m["F"], m["lowest_elevation_angle"], m["highest_elevation_angle"],
                m["inner_img_radius"], m["outer_img_radius"], center_point=m["center_point"],
                image_size=tuple(m["image_size"]), units=d["units"], center_point_inner=m.get("center_point_inner"),
                center_point_outer=m.get("center_point_outer"))   # (optional keys: older documents have one centre)
        if m.get("R_model_wrt_C") is not None:   # orientation of the mirror frame wrt [C] (camera_models.py:955-962); default identity
            g.set_pose(g.F[:3, 0], np.asarray(m["R_model_wrt_C"], dtype=np.float64))
        return g
    gums = GUMStereo(mirror(d["top"]), mirror(d["bottom"]), units=d["units"])
    if with_panoramas:
        for m in (gums.top_model, gums.bot_model):
            m.panorama = Panorama(m, width=int(d.get("panorama_width", 1200)))
    return gums


def save_gums_json(gums, filename):
    import json
    with open(filename, "w") as f:
        json.dump(gums_to_dict(gums), f, indent=1)


def load_gums_json(filename, with_panoramas=True):
    import json
    with open(filename) as f:
        return gums_from_dict(json.load(f), with_panoramas=with_panoramas)
