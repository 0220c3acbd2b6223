"""Host-side mirror of the hot-path part of omnistereo/pose_est_tools.py: the per-frame classes and functions of
one VO step with the reference's names, arguments, return structure and error behaviour; the arithmetic runs
on the GPU through libsosvo (FeatureMatcher, pyopengv mirror, the rig's device front end).

    get_selected_distances_to_model / select_inliers_within_distance   pose_est_tools.py:131-203  (numpy, as there)
    match_features_frame_to_frame                                       pose_est_tools.py:211-269
    StereoPanoramicFrame                                                pose_est_tools.py:271-402
    TrackerSE3 / TrackerStereoSE3 (track_frame, bootstrap_tracker)      pose_est_tools.py:594-878

The batched throughput path (vo_single_camera_sos_amd.pipeline.FramePairPipeline) runs the same sequence for many
frame pairs at once without host round trips; this module is the drop-in for code written against the
reference's per-frame API.  Out of scope here: run_VO, keyframe policy, TUM writer,
visualisation, the live-camera driver."""
from math import log10, sqrt

import numpy as np

from .. import pyopengv
from .camera_models import FeatureMatcher, PanoramicCorrespondences, keypoints_to_array
from .common_cv import filter_pixel_correspondences


def get_length_units_conversion_factor(input_units, output_units):
    """common_tools.py:580-597"""
    table = {("cm", "mm"): 10.0, ("cm", "m"): 0.01, ("mm", "cm"): 0.1, ("mm", "m"): 0.001, ("m", "mm"): 1000.0,
             ("m", "cm"): 100.0}
    return table.get((input_units, output_units), 1.0)


def normalized(v):
    return v / np.linalg.norm(v)


def get_selected_distances_to_model(model, indices, bearing_vectors1, bearing_vectors2, is_relative_2D_to_2D_case,
                                    verbose_debug=False):
    """pose_est_tools.py:150-203: OpenGV's absolute-pose score 1 - f . normalize(R^T (p - t)) per index (the
    relative 2D-2D case adds the score of the triangulated point in frame 1).  Unused at run time in the
    reference too; it pins the scoring definition the RANSAC kernel implements."""
    translation = model[:3, 3]
    rotation = model[:3, :3]
    inverse = np.identity(4)
    inverse[:3, :3] = rotation.T
    inverse[:3, 3] = -inverse[:3, :3].dot(translation[:3])
    if is_relative_2D_to_2D_case:
        p_all = pyopengv.triangulation_triangulate2(bearing_vectors1[..., :3], bearing_vectors2[..., :3], translation, rotation)
    else:
        p_all = bearing_vectors1.copy()
    scores = []
    for i in indices:
        p_homo = np.ones(4)
        p_homo[:3] = p_all[i][:3]
        reprojection2 = normalized(inverse.dot(p_homo)[:3])
        score = 1.0 - bearing_vectors2[i, :3].T.dot(reprojection2)
        if is_relative_2D_to_2D_case:
            score = (1.0 - bearing_vectors1[i, :3].T.dot(normalized(p_homo[:3]))) + score
        scores.append(score)
    return scores


def select_inliers_within_distance(model_coefficients, indices_all, threshold, bearing_vectors1, bearing_vectors2,
                                   is_relative_2D_to_2D_case):
    """pose_est_tools.py:131-148"""
    d = np.array(get_selected_distances_to_model(model_coefficients, indices_all, bearing_vectors1, bearing_vectors2,
                                                 is_relative_2D_to_2D_case))
    test = d < threshold
    return indices_all[test], indices_all[np.invert(test)]


def match_features_frame_to_frame(cam_model, train_kpts, train_desc, query_kpts, query_desc, random_colors_RGB,
                                  max_horizontal_diff=-1, max_descriptor_distance_radius=-1, keypts_as_points_train=None,
                                  keypts_as_points_query=None, pano_img_train=None, pano_img_query=None,
                                  show_matches=False, win_name="Matches (Frame-to-Frame)"):
    """pose_est_tools.py:211-269: query = current frame, train = keyframe; first percentage_good_matches of the
    sorted matches; gate |du| <= max_horizontal_diff (no wrap-around, as the reference).
    -> (train idx, train kpts, train desc), (query idx, query kpts, query desc), colours"""
    fm = cam_model.feature_matcher_for_motion
    q, t, _ = fm.match_arrays(query_desc, train_desc)
    matched_train_indices, matched_query_indices = [], []
    matched_kpts_train, matched_kpts_query, matched_desc_train, matched_desc_query, random_colors = [], [], [], [], []
    good = int(fm.percentage_good_matches * len(q))
    if good > 0:
        t_all, q_all = t[:good], q[:good]
        train_kpts, query_kpts = np.asarray(train_kpts, dtype=object), np.asarray(query_kpts, dtype=object)
        k_train_all, k_query_all = train_kpts[t_all], query_kpts[q_all]
        if keypts_as_points_train is None:
            keypts_as_points_train = keypoints_to_array(k_train_all).astype(np.float64)
        else:
            keypts_as_points_train = np.asarray(keypts_as_points_train)[t_all]
        if keypts_as_points_query is None:
            keypts_as_points_query = keypoints_to_array(k_query_all).astype(np.float64)
        else:
            keypts_as_points_query = np.asarray(keypts_as_points_query)[q_all]
        if max_horizontal_diff >= 0:
            ok = filter_pixel_correspondences(matched_points_top=keypts_as_points_train,
                                              matched_points_bot=keypts_as_points_query, min_rectified_disparity=-1,
                                              max_horizontal_diff=max_horizontal_diff)
            matched_train_indices, matched_query_indices = t_all[ok], q_all[ok]
            matched_kpts_train, matched_kpts_query = k_train_all[ok], k_query_all[ok]
        else:
            matched_train_indices, matched_query_indices = t_all, q_all
            matched_kpts_train, matched_kpts_query = k_train_all, k_query_all
        matched_desc_train = np.asarray(train_desc)[matched_train_indices]
        matched_desc_query = np.asarray(query_desc)[matched_query_indices]
        try:
            random_colors = np.asarray(random_colors_RGB)[matched_train_indices]
        except Exception:
            print("Problem, it only has", len(random_colors_RGB))
    return (matched_train_indices, matched_kpts_train, matched_desc_train), \
        (matched_query_indices, matched_kpts_query, matched_desc_query), random_colors


class StereoPanoramicFrame(object):
    """pose_est_tools.py:271-402: one omnistereo frame = detected + stereo-matched + triangulated + range-filtered
    correspondences of the rig's current omni image."""

    def __init__(self, stereo_camera_model, frame_id, **kwargs):
        self.frame_id = frame_id
        self.parent_id = kwargs.get("parent_id", -1)
        self.T_frame_wrt_tracking_ref_frame = np.identity(4)
        top, bot = stereo_camera_model.top_model, stereo_camera_model.bot_model
        self.panoramic_image_top = None if top.panorama.panoramic_img is None else top.panorama.panoramic_img.copy()
        self.panoramic_image_bottom = None if bot.panorama.panoramic_img is None else bot.panorama.panoramic_img.copy()
        self.use_midpoint_triangulation = True
        self.use_opengv_triangulation = False
        self.conversion_factor_length_to_m = get_length_units_conversion_factor(stereo_camera_model.units, "m")
        self.first_row_to_crop_bottom = 0
        self.total_time = 0.
        self.median_win_size = 11
        self.min_disp = 1
        self.max_u_dist = 2.5 if self.use_midpoint_triangulation else 0.5
        from_m = get_length_units_conversion_factor("m", stereo_camera_model.units)
        self.min_range = 0.5 * from_m
        self.max_range = 7.0 * from_m
        self.pano_correspondences = None
        self.num_valid_keypoints = 0
        self.establish_stereo_correspondences(omnistereo_model=stereo_camera_model)

    def establish_stereo_correspondences(self, omnistereo_model, collect_time_statistics=False):
        fm = omnistereo_model.feature_matcher_for_static_stereo
        top, bot = omnistereo_model.top_model, omnistereo_model.bot_model
        kl_top, dl_top = top.detect_sparse_features_on_panorama(feature_detection_method=fm.feature_detection_method,
                                                                num_of_features=fm.num_of_features,
                                                                median_win_size=self.median_win_size, show=False)
        kl_bot, dl_bot = bot.detect_sparse_features_on_panorama(feature_detection_method=fm.feature_detection_method,
                                                                num_of_features=fm.num_of_features,
                                                                median_win_size=self.median_win_size, show=False)
        (m_top0, k_top0, d_top0), (m_bot0, k_bot0, d_bot0), colors0 = omnistereo_model.match_features_panoramic_top_bottom(
            keypts_list_top=kl_top, desc_list_top=dl_top, keypts_list_bot=kl_bot, desc_list_bot=dl_bot,
            min_rectified_disparity=self.min_disp, max_horizontal_diff=self.max_u_dist, show_matches=False)
        az1, el1 = top.panorama.get_direction_angles_from_pixel_pano(m_top0, use_LUTs=False)
        az2, el2 = bot.panorama.get_direction_angles_from_pixel_pano(m_bot0, use_LUTs=False)
        b1 = top.get_3D_point_from_angles_wrt_focus(azimuth=az1, elevation=el1)[0, ..., :3]
        b2 = bot.get_3D_point_from_angles_wrt_focus(azimuth=az2, elevation=el2)[0, ..., :3]
        if self.use_opengv_triangulation:
            T = omnistereo_model.T_bot_wrt_top
            xyz_top = pyopengv.triangulation_triangulate2(b1, b2, T[:3, 3], T[:3, :3])
            xyz_top_homo = np.concatenate((xyz_top, np.ones(xyz_top.shape[:-1])[..., np.newaxis]), axis=-1)
            xyz0 = np.einsum("ij, nj->ni", top.T_model_wrt_C, xyz_top_homo)
        else:
            xyz0 = omnistereo_model.get_triangulated_point_from_direction_angles(
                dir_angs_top=(az1, el1), dir_angs_bot=(az2, el2), use_midpoint_triangulation=self.use_midpoint_triangulation)[0]
        good = omnistereo_model.filter_panoramic_points_due_to_range(xyz0, min_3D_range=self.min_range,
                                                                     max_3D_range=self.max_range)
        self.num_valid_keypoints = int(np.count_nonzero(good))
        self.bearing_vectors_top_stereo_triangulated = b1[good]
        self.bearing_vectors_bottom_stereo_triangulated = b2[good]
        self.pano_correspondences = PanoramicCorrespondences(
            kpts_top_list=k_top0[good], desc_top_list=d_top0[good], kpts_bot_list=k_bot0[good], desc_bot_list=d_bot0[good],
            points_3D=xyz0[good], m_top_array=m_top0[good], m_bot_array=m_bot0[good],
            random_colors_RGB_list=colors0[good], do_flattening=False)


class TrackerSE3(object):
    """pose_est_tools.py:594-720 (the tracking parameters; visualisation and result paths left out)."""

    def __init__(self, camera_model, show_3D_points=False, **kwargs):
        self.camera_model = camera_model
        self.show_3D_points = show_3D_points
        self.T_C_wrt_S_init = np.identity(4)
        self.T_C_curr_frame_wrt_S_est = np.identity(4)
        self.xyz_homo_points_wrt_C_inliers = []
        self.rgb_points_inliers = []
        self.num_tracked_correspondences = 0
        self.inlier_tracked_correspondences_ratio = 0.
        self.number_of_cams = 1
        self.T_Ckey_wrt_S_est_list = []
        self.set_global_parameters_for_tracking()

    def set_global_parameters_for_tracking(self):
        """pose_est_tools.py:672-707"""
        self.backprojection_score_threshold_3D_to_2D_in_degrees = 5.
        self.backprojection_score_threshold_3D_to_2D = 1.0 - np.cos(np.deg2rad(self.backprojection_score_threshold_3D_to_2D_in_degrees))
        self.detection_method = "GFT"
        self.matching_type = "BF"
        self.k_best_matches = 1
        self.percentage_good_matches = 1.0
        self.use_descriptor_radius_match_for_motion = False
        self.num_features_detection_for_motion = 1000
        self.max_horizontal_search_ratio = 0.50
        self.pose_est_algorithm = "EPNP"
        self.n_points_for_RANSAC_model = 3
        self.correspondences_outliers_fraction = 0.65
        self.max_ransac_iterations_3D_to_2D = -1
        if self.max_ransac_iterations_3D_to_2D < 0:
            self.max_ransac_iterations_3D_to_2D = self.compute_num_of_iterations_RANSAC(
                n_points_for_model=self.n_points_for_RANSAC_model,
                correspondences_outliers_fraction=self.correspondences_outliers_fraction)

    def compute_num_of_iterations_RANSAC(self, n_points_for_model, correspondences_outliers_fraction):
        """pose_est_tools.py:709-720"""
        w = 1.0 - correspondences_outliers_fraction
        desired_prob_only_inlier_selection = 0.998
        num_of_iters = log10(1.0 - desired_prob_only_inlier_selection) / log10(1.0 - w ** n_points_for_model)
        std_of_k = sqrt(1.0 - w ** n_points_for_model) / (w ** n_points_for_model)
        return int(num_of_iters + 3 * std_of_k)


class TrackerStereoSE3(TrackerSE3):
    """pose_est_tools.py:722-878"""

    def __init__(self, camera_model, show_3D_points=False, **kwargs):
        TrackerSE3.__init__(self, camera_model, show_3D_points, **kwargs)
        self.omnistereo_model = self.camera_model
        self.number_of_cams = 2
        self.bootstrap_tracker()

    def bootstrap_tracker(self):
        """pose_est_tools.py:849-878"""
        om = self.omnistereo_model
        self.cam_offsets = np.array([om.top_model.T_model_wrt_C[:3, 3], om.bot_model.T_model_wrt_C[:3, 3]])
        self.cam_rotations = np.array([om.top_model.T_model_wrt_C[:3, :3], om.bot_model.T_model_wrt_C[:3, :3]])
        self.num_features_detection_for_static_stereo = 1000
        om.feature_matcher_for_static_stereo = FeatureMatcher(
            method=self.detection_method, matcher_type=self.matching_type, k_best=self.k_best_matches,
            percentage_good_matches=self.percentage_good_matches, num_of_features=self.num_features_detection_for_static_stereo,
            use_radius_match=False)
        self.max_horizontal_diff_f2f_matches = 0.125 * self.max_horizontal_search_ratio * om.top_model.panorama.cols
        om.feature_matcher_for_motion = FeatureMatcher(
            method=self.detection_method, matcher_type=self.matching_type, k_best=self.k_best_matches,
            percentage_good_matches=self.percentage_good_matches, num_of_features=self.num_features_detection_for_motion,
            use_radius_match=self.use_descriptor_radius_match_for_motion)
        self.omni_mask_extra_padding = 10
        shape = None if om.current_omni_img is None else om.current_omni_img.shape[:2]
        if shape is None:
            shape = (om.top_model.image_size[1], om.top_model.image_size[0])
        if om.top_model.mask is None or om.bot_model.mask is None:
            om.make_annulus_masks(shape)
        for m in (om.top_model, om.bot_model):
            m.panorama.generate_azimuthal_masks(azimuth_mask_degrees=30, overlap_degrees=0, show=False,
                                                elev_mask_padding=self.omni_mask_extra_padding,
                                                stand_masks_azimuth_coord_in_degrees_list=[50, 170, 290],
                                                stand_masks_width_in_degrees=10, omni_shape=shape)

    def track_frame(self, reference_frame, current_frame):
        """pose_est_tools.py:736-847 -> (ok, message); sets current_frame.T_frame_wrt_tracking_ref_frame and
        self.T_C_curr_frame_wrt_S_est."""
        self.num_tracked_correspondences = 0
        self.inlier_tracked_correspondences_ratio = 0.
        ref, cur = reference_frame.pano_correspondences, current_frame.pano_correspondences
        (t_top, _, _), (q_top, _, _), _ = match_features_frame_to_frame(
            cam_model=self.omnistereo_model, train_kpts=ref.kpts_top, train_desc=ref.desc_top, query_kpts=cur.kpts_top,
            query_desc=cur.desc_top, random_colors_RGB=ref.random_colors_RGB,
            max_horizontal_diff=self.max_horizontal_diff_f2f_matches, max_descriptor_distance_radius=-1,
            keypts_as_points_train=ref.m_top, keypts_as_points_query=cur.m_top)
        (t_bot, _, _), (q_bot, _, _), _ = match_features_frame_to_frame(
            cam_model=self.omnistereo_model, train_kpts=ref.kpts_bot, train_desc=ref.desc_bot, query_kpts=cur.kpts_bot,
            query_desc=cur.desc_bot, random_colors_RGB=ref.random_colors_RGB,
            max_horizontal_diff=self.max_horizontal_diff_f2f_matches, max_descriptor_distance_radius=-1,
            keypts_as_points_train=ref.m_bot, keypts_as_points_query=cur.m_bot)
        t_top, q_top, t_bot, q_bot = [np.asarray(a, dtype=np.int64) for a in (t_top, q_top, t_bot, q_bot)]
        bearings = [current_frame.bearing_vectors_top_stereo_triangulated[q_top],
                    current_frame.bearing_vectors_bottom_stereo_triangulated[q_bot]]
        points = [ref.points_3D_coords_homo[t_top][..., :3], ref.points_3D_coords_homo[t_bot][..., :3]]
        cam_all = np.concatenate([np.zeros((len(points[c]), 1)) + float(c) for c in range(self.number_of_cams)])
        b_all = np.vstack(bearings).reshape(-1, 3)
        p_all = np.vstack(points).reshape(-1, 3)
        num_initial_matches = b_all.shape[0]
        if num_initial_matches < 2 * self.n_points_for_RANSAC_model * (0.33 * self.number_of_cams):
            return False, "Cannot track on only %d point correspondences" % (num_initial_matches)
        T_ransac, inliers = pyopengv.absolute_pose_noncentral_ransac(
            b_all, cam_all, p_all, self.cam_offsets, self.cam_rotations, self.backprojection_score_threshold_3D_to_2D,
            self.max_ransac_iterations_3D_to_2D)
        self.indices_inliers_combined = inliers
        self.num_tracked_correspondences = len(inliers)
        self.inlier_tracked_correspondences_ratio = float(self.num_tracked_correspondences) / float(num_initial_matches)
        T_nl = pyopengv.absolute_pose_noncentral_optimize_nonlinear(
            b_all[inliers], cam_all[inliers], p_all[inliers], self.cam_offsets, self.cam_rotations, T_ransac[:3, 3],
            T_ransac[:3, :3])
        T_homo = np.identity(4)
        T_homo[:3] = T_nl
        T_homo[:3, 3] = T_nl[:3, 3] * current_frame.conversion_factor_length_to_m
        current_frame.T_frame_wrt_tracking_ref_frame = T_homo
        T_key = self.T_Ckey_wrt_S_est_list[-1] if self.T_Ckey_wrt_S_est_list else np.identity(4)
        self.T_C_curr_frame_wrt_S_est = T_key.dot(T_homo)
        if self.show_3D_points:
            self.xyz_homo_points_wrt_C_inliers = np.hstack((p_all[inliers] * current_frame.conversion_factor_length_to_m,
                                                            np.ones((len(inliers), 1))))
        return True, "tracking used %d inlier point correspondences" % (self.num_tracked_correspondences)
