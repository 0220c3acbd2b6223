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


class GUM(object):
    """One mirror.  `F` is the focus (viewpoint) wrt the common frame [C]; `z_axis` is +1 for the top
    mirror, -1 for the bottom one (gum.py:335-341)."""

    def __init__(self, params, z_axis, F, lowest_elevation_angle, highest_elevation_angle, inner_img_radius,
                 outer_img_radius, center_point=None, image_size=(640, 480), mirror_name="", units="mm"):
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
    @staticmethod
    def _disc(shape, center, radius):
        yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
        return (xx - int(center[0])) ** 2 + (yy - int(center[1])) ** 2 <= int(radius) ** 2

    def make_mask(self, mask_shape, radius_pixel_shrinking=0):
        r_in = self.inner_img_radius + radius_pixel_shrinking
        r_out = self.outer_img_radius - radius_pixel_shrinking
        m = self._disc(mask_shape, self.center_point, r_out)
        if r_in > 0:
            m &= ~self._disc(mask_shape, self.center_point, r_in)
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

    def get_baseline(self):
        return self.top_model.F[2, 0] - self.bot_model.F[2, 0]

    def make_annulus_masks(self, shape):
        """camera_models.py:2944-2988: top = outer disc minus inner disc minus the bottom mirror's outer disc;
        bottom = its outer disc intersected with the top mirror's inner disc, minus its own inner disc."""
        t, b = self.top_model, self.bot_model
        top = GUM._disc(shape, t.center_point, t.outer_img_radius)
        if t.inner_img_radius > 0:
            top &= ~GUM._disc(shape, t.center_point, t.inner_img_radius)
            if b.outer_img_radius > 0:
                top &= ~GUM._disc(shape, b.center_point, b.outer_img_radius)
        bot = GUM._disc(shape, b.center_point, b.outer_img_radius) & GUM._disc(shape, b.center_point, t.inner_img_radius)
        bot &= ~GUM._disc(shape, b.center_point, b.inner_img_radius)
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
