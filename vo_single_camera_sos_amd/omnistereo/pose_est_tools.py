"""Host-side mirror of the hot-path part of omnistereo/pose_est_tools.py: the per-frame classes and functions of
one VO step with the reference's names, arguments, return structure and error behaviour; the arithmetic runs
on the GPU through libsosvo (FeatureMatcher, pyopengv mirror, the rig's device front end).

    get_selected_distances_to_model / select_inliers_within_distance   pose_est_tools.py:131-203  (numpy, as there)
    match_features_frame_to_frame                                       pose_est_tools.py:211-269
    StereoPanoramicFrame                                                pose_est_tools.py:271-402
    TrackerSE3 / TrackerStereoSE3 (track_frame, bootstrap_tracker)      pose_est_tools.py:594-878

The batched throughput path (vo_single_camera_sos_amd.pipeline.FramePairPipeline) runs the same sequence for many
frame pairs at once without host round trips; this module is the drop-in for code written against the
reference's per-frame API.

    RGBDFrame / TrackerRGBDSE3                                          pose_est_tools.py:404-623, :880-958
    StereoPanoramicKeyFrame / RGBDKeyFrame                              pose_est_tools.py:625-641
    run_VO / driver_VO (frame loop, keyframe policy, TUM pose files)    pose_est_tools.py:1264-1741

Not built: visualisation (vispy / matplotlib windows), the live-camera driver (run_VO_live)."""
import os
import threading
import time
from warnings import warn

from math import log10, sqrt

import numpy as np

from .. import pyopengv
from . import transformations as tr
from .camera_models import FeatureMatcher, KeyPoint, KeyPointAndDescriptor, PanoramicCorrespondences, keypoints_to_array
from .common_cv import filter_pixel_correspondences
from .common_tools import copy_only_attributes, get_length_units_conversion_factor, make_sure_path_exists

_IDENTITY4 = np.identity(4)
_IDENTITY4.setflags(write=False)   # shared initial pose of device frames: replaced by the tracker, never written in place


def normalized(v):
    return v / np.linalg.norm(v)


def pose_relative_ransac_2D_to_2D(bearing_vectors1, bearing_vectors2, model_error_threshold=0.001,
                                  rel_pose_est_algorithm="STEWENIUS", outlier_fraction_known=0.50):
    """pose_est_tools.py:54-90: 2D-2D relative-pose RANSAC with the iteration budget N = log(0.01) / log(1 - w^n) + 3 std,
    n = 5 ("NISTER", "STEWENIUS"), 7 ("SEVENPT") or 8 ("EIGHTPT") -> (T 4x4 with |t| = 1, inlier indices).  Not reached
    by the VO drivers."""
    from math import log10, sqrt
    n_points_for_model = {"NISTER": 5, "STEWENIUS": 5, "SEVENPT": 7, "EIGHTPT": 8}.get(rel_pose_est_algorithm, -1)
    w = 1.0 - outlier_fraction_known
    num_of_iters = log10(1.0 - 0.99) / log10(1.0 - w ** n_points_for_model)
    std_of_k = sqrt(1.0 - w ** n_points_for_model) / (w ** n_points_for_model)
    max_iterations = int(num_of_iters + 3 * std_of_k)
    T, inliers = pyopengv.relative_pose_ransac(bearing_vectors1[..., :3], bearing_vectors2[..., :3], rel_pose_est_algorithm,
                                               model_error_threshold, max_iterations)
    T_homo = np.identity(4)
    T_homo[:3] = T
    return T_homo, inliers


def pose_absolute_ransac_3D_to_2D(bearing_vectors, points3D, model_error_threshold=0.001, pose_est_algorithm="EPNP",
                                  outlier_fraction_known=0.50, max_iterations=-1):
    """pose_est_tools.py:92-129: central absolute-pose RANSAC with the iteration budget N = log(0.01) / log(1 - w^n)
    + 3 std (n = 2 for "TWOPT", else 3 -- as written there, also for "EPNP") when max_iterations < 0
    -> (T 4x4, inlier indices)."""
    from math import log10, sqrt
    if max_iterations < 0:
        n_points_for_model = 2 if pose_est_algorithm == "TWOPT" else 3
        w = 1.0 - outlier_fraction_known
        num_of_iters = log10(1.0 - 0.99) / log10(1.0 - w ** n_points_for_model)
        std_of_k = sqrt(1.0 - w ** n_points_for_model) / (w ** n_points_for_model)
        max_iterations = int(num_of_iters + 3 * std_of_k)
    T, inliers = pyopengv.absolute_pose_ransac(bearing_vectors[..., :3], points3D[..., :3], pose_est_algorithm,
                                               model_error_threshold, max_iterations)
    T_homo = np.identity(4)
    T_homo[:3] = T
    return T_homo, inliers


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
        if isinstance(current_frame, DeviceStereoFrame):
            return self._track_device_frame(reference_frame, current_frame)
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


    def _track_device_frame(self, reference_frame, current_frame):
        """track_frame on frames of a SequenceEngine's store: the [16] record (refined pose, inliers, correspondences,
        status) of the pair -- the speculative one when the reference is the frame's predecessor, one serial tracking
        call against the keyframe slot otherwise -- put through the same bookkeeping as above."""
        eng = current_frame.engine
        ref_seq = getattr(reference_frame, "seq_index", None)
        if current_frame.spec_record is not None and ref_seq == current_frame.seq_index - 1:
            rec = current_frame.spec_record
        elif current_frame.spec2_record is not None and ref_seq == current_frame.seq_index - 2:
            rec = current_frame.spec2_record      # the frame in between was not promoted: tracked against t - 2 in the batch
        else:
            rec = eng.track(eng.key_slot, current_frame.slot, current_frame.seed)
        num_initial_matches = int(rec[13])
        if num_initial_matches < 2 * self.n_points_for_RANSAC_model * (0.33 * self.number_of_cams):
            return False, "Cannot track on only %d point correspondences" % (num_initial_matches)
        self.indices_inliers_combined = None    # (the inlier list stays on the device)
        self.num_tracked_correspondences = int(rec[12])
        self.inlier_tracked_correspondences_ratio = float(self.num_tracked_correspondences) / float(num_initial_matches)
        T_homo = np.identity(4)
        T_homo[:3] = rec[:12].reshape(3, 4)
        T_homo[:3, 3] = T_homo[:3, 3] * current_frame.conversion_factor_length_to_m
        current_frame.T_frame_wrt_tracking_ref_frame = T_homo
        T_key = self.T_Ckey_wrt_S_est_list[-1] if self.T_Ckey_wrt_S_est_list else np.identity(4)
        self.T_C_curr_frame_wrt_S_est = T_key.dot(T_homo)
        return True, "tracking used %d inlier point correspondences" % (self.num_tracked_correspondences)


class StereoPanoramicKeyFrame(StereoPanoramicFrame):
    """pose_est_tools.py:625-632: a frame promoted to keyframe (a blind copy of its attributes)."""

    def __init__(self, frame, **kwargs):
        if isinstance(frame, DeviceStereoFrame):   # (plain instance attributes only: the same blind copy, without dir())
            self.__dict__.update(frame.__dict__)
        else:
            copy_only_attributes(objfrom=frame, objto=self)
        self.children_ids = []


class DeviceStereoFrame(StereoPanoramicFrame):
    """A StereoPanoramicFrame whose correspondences live in the frame store of a pipeline.SequenceEngine (HBM) instead
    of host arrays: run_VO's sequence mode builds these, `window` frames per batched front-end pass.  Same attributes as
    the host frame except pano_correspondences / the bearing arrays, which stay on the device (slot `slot` of the
    store while the frame's window is current; the keyframe slot once promoted)."""

    def __init__(self, engine, info, stereo_camera_model, frame_id, seq_index, **kwargs):
        consts = getattr(engine, "_frame_consts", None)
        if consts is None:   # the same for every frame of the engine (ctypes field reads are slow: once)
            consts = engine._frame_consts = dict(
                panoramic_image_top=None, panoramic_image_bottom=None, use_midpoint_triangulation=True,
                use_opengv_triangulation=False, first_row_to_crop_bottom=0, total_time=0., pano_correspondences=None,
                conversion_factor_length_to_m=get_length_units_conversion_factor(stereo_camera_model.units, "m"),
                median_win_size=int(engine.cfg.median_ksize), min_disp=engine.rig_cfg.stereo_min_disp,
                max_u_dist=engine.rig_cfg.stereo_max_hdiff, min_range=engine.rig_cfg.min_range, max_range=engine.rig_cfg.max_range)
        self.__dict__.update(consts)
        self.frame_id = frame_id
        self.parent_id = kwargs.get("parent_id", -1)
        self.T_frame_wrt_tracking_ref_frame = _IDENTITY4   # (replaced, never written in place, by the tracker)
        self.engine, self.slot, self.seq_index = engine, int(info["slot"]), int(seq_index)
        self.seed, self.spec_record, self.spec2_record = int(info["seed"]), info["spec"], info.get("spec2")
        self.num_valid_keypoints = int(info["count"])

    def promote(self):
        self.engine.promote(self.slot)


def sequence_engine_for(tracker, camera_model, first_image, window):
    """The pipeline.SequenceEngine that computes what `tracker` (a TrackerStereoSE3 with the reference's settings) and
    StereoPanoramicFrame compute, or None when the configuration has no batched counterpart (then run_VO keeps the
    per-frame mirror path)."""
    from ..pipeline import RigConfig, SequenceEngine
    om = camera_model
    if (tracker.detection_method != "GFT" or tracker.matching_type != "BF" or tracker.k_best_matches != 1
            or tracker.use_descriptor_radius_match_for_motion):
        return None
    top, bot = om.top_model, om.bot_model
    if not (np.array_equal(tracker.cam_rotations[0], np.identity(3)) and np.array_equal(tracker.cam_rotations[1], np.identity(3))
            and np.array_equal(tracker.cam_offsets[0], top.F[:3, 0]) and np.array_equal(tracker.cam_offsets[1], bot.F[:3, 0])):
        return None
    if om.current_omni_img is None:
        om.current_omni_img = first_image
    model = om._device_model()
    from_m = get_length_units_conversion_factor("m", om.units)
    geo = [(m.panorama.cols, m.panorama.rows, m.panorama.pixel_size, m.panorama.cyl_height_max) for m in (top, bot)]
    rig = RigConfig(pano_top=geo[0], pano_bot=geo[1], F_top=top.F[:3, 0], F_bot=bot.F[:3, 0], min_range=0.5 * from_m,
                    max_range=7.0 * from_m, stereo_min_disp=1.0, stereo_max_hdiff=2.5,
                    f2f_max_hdiff=tracker.max_horizontal_diff_f2f_matches, pct_good_matches=tracker.percentage_good_matches)
    nfeat = tracker.num_features_detection_for_static_stereo
    cap = int(min(4096, max(64, -(-int(nfeat) // 64) * 64)))      # as GUMStereo._front_end (the mirror path's capacity)
    # an engine (frame store, pinned staging buffers, scratch) is kept with the rig and reused by later runs with the same
    # settings: building one costs more than tracking a hundred frames
    key = (int(window), int(nfeat), cap, float(tracker.backprojection_score_threshold_3D_to_2D),
           int(tracker.max_ransac_iterations_3D_to_2D), float(tracker.max_horizontal_diff_f2f_matches),
           float(tracker.percentage_good_matches), tuple(geo[0]), tuple(geo[1]), tuple(top.F[:3, 0]), tuple(bot.F[:3, 0]), id(model))
    cache = om.__dict__.setdefault("_sequence_engines", {})
    eng = cache.get(key)
    if eng is None:
        cache.clear()   # (one engine per rig: a new configuration replaces the old one's buffers)
        eng = cache[key] = SequenceEngine(om._context(), model, rig, window=window, num_of_features=nfeat, kp_cap=cap,
                                          frame_cap=4096, median_win_size=11, thr=tracker.backprojection_score_threshold_3D_to_2D,
                                          max_iter=tracker.max_ransac_iterations_3D_to_2D, adaptive=True,
                                          lm_iter=pyopengv.LM_MAX_ITERATIONS, ransac_solver="GP3P")
    else:
        eng.reset()
    return eng


class RGBDFrame(object):
    """pose_est_tools.py:404-623: one RGB-D frame = keypoints with valid depth, their descriptors, 3-D points and
    bearings.  Detection + description + back-projection run on the GPU (RGBDFrontEnd, one frame)."""

    def __init__(self, rgbd_camera_model, frame_id, **kwargs):
        self.rgbd_camera_model = rgbd_camera_model
        self.frame_id = frame_id
        self.parent_id = kwargs.get("parent_id", -1)
        self.T_frame_wrt_tracking_ref_frame = kwargs.get("T_wrt_ref", np.identity(4))
        self.conversion_factor_length_to_m = get_length_units_conversion_factor(rgbd_camera_model.units, "m")
        self.total_time = 0.
        self.median_win_size = 0
        self.min_range = 0.8
        self.max_range = 7.
        self.mask = kwargs.get("mask", None)
        self.rgb_img = kwargs.get("rgb_img", None)
        self.depth_map = kwargs.get("depth_map", None)
        self.num_valid_keypoints = 0
        self.keypoints_and_descriptors = None
        self.bearing_vectors = None
        self.keypoints_3D_points = None
        self.current_depth = None
        if (self.rgb_img is not None) and (self.depth_map is not None):
            self.establish_keypoints(rgb=self.rgb_img, depth=self.depth_map)

    def filter_3D_points_due_to_range(self, xyz_points_wrt_C, min_3D_range=0, max_3D_range=0.):
        """pose_est_tools.py:570-592 (numpy, as there)."""
        valid = np.ones(shape=(xyz_points_wrt_C.shape), dtype="bool")
        if min_3D_range > 0 or max_3D_range > 0:
            nrm = np.nan_to_num(np.linalg.norm(xyz_points_wrt_C, axis=0, keepdims=True), copy=True)
            if min_3D_range > 0:
                valid = np.logical_and(valid, nrm >= min_3D_range)
            if max_3D_range > 0:
                valid = np.logical_and(valid, nrm <= max_3D_range)
        return valid

    def establish_keypoints(self, rgb, depth):
        """pose_est_tools.py:600-623.  `rgb` is handed to the detector as it is (the gray conversion weights the
        first channel as blue, like cv2.COLOR_BGR2GRAY applied to whatever the caller passes, :531)."""
        fm = self.rgbd_camera_model.feature_matcher_for_motion
        method = "GFT" if fm is None else str(fm.feature_detection_method).upper()
        if method != "GFT":
            raise NotImplementedError("RGB-D detection method %r: GFT (the trackers' default, pose_est_tools.py:684) is built"
                                      % method)
        nfeat = 50 if fm is None else fm.num_of_features
        self.current_depth = depth
        fe = self.rgbd_camera_model._front_end(np.asarray(rgb).shape[:2], nfeat, self.median_win_size, self.min_range,
                                               self.max_range, self.mask)
        fe.load_frames(np.ascontiguousarray(rgb)[None], np.ascontiguousarray(depth, dtype=np.float32)[None])
        fe.run()
        fe.ctx.synchronize()
        M = int(fe.frames["M"][0])
        m = fe.frames["m"][0, :M].cpu().numpy()
        self.keypoints_3D_points = fe.frames["X"][0, :M].cpu().numpy()
        self.bearing_vectors = fe.frames["b"][0, :M].cpu().numpy()
        kpts = [KeyPoint(x, y, size=31.0) for x, y in m]
        self.keypoints_and_descriptors = KeyPointAndDescriptor(kpts_list=kpts, desc_list=fe.frames["d"][0, :M].cpu().numpy(),
                                                               coords_array=m.astype(np.float64), do_flattening=False)
        self.num_valid_keypoints = M


class RGBDKeyFrame(RGBDFrame):
    """pose_est_tools.py:634-641"""

    def __init__(self, frame, **kwargs):
        if isinstance(frame, DeviceRGBDFrame):   # (plain instance attributes only: the same blind copy, without dir())
            self.__dict__.update(frame.__dict__)
        else:
            copy_only_attributes(objfrom=frame, objto=self)
        self.children_ids = []


class DeviceRGBDFrame(RGBDFrame):
    """An RGBDFrame whose keypoints, descriptors, points and bearings live in the frame store of a
    pipeline.RGBDSequenceEngine (HBM): run_VO's sequence mode builds these, `window` frames per batched front-end pass."""

    def __init__(self, engine, info, rgbd_camera_model, frame_id, seq_index, **kwargs):
        self.rgbd_camera_model = rgbd_camera_model
        self.frame_id = frame_id
        self.parent_id = kwargs.get("parent_id", -1)
        self.T_frame_wrt_tracking_ref_frame = np.identity(4)
        self.conversion_factor_length_to_m = get_length_units_conversion_factor(rgbd_camera_model.units, "m")
        self.total_time = 0.
        self.median_win_size = engine.cfg.median_ksize
        self.min_range, self.max_range = engine.cam_cfg.min_range, engine.cam_cfg.max_range
        self.mask = None
        self.rgb_img = self.depth_map = self.current_depth = None
        self.keypoints_and_descriptors = self.bearing_vectors = self.keypoints_3D_points = None
        self.engine, self.slot, self.seq_index = engine, int(info["slot"]), int(seq_index)
        self.seed, self.spec_record, self.spec2_record = int(info["seed"]), info["spec"], info.get("spec2")
        self.num_valid_keypoints = int(info["count"])

    def promote(self):
        self.engine.promote(self.slot)


def rgbd_sequence_engine_for(tracker, camera_model, first_rgb, window):
    """The pipeline.RGBDSequenceEngine that computes what `tracker` (a TrackerRGBDSE3) and RGBDFrame compute, or None when
    the configuration has no batched counterpart (then run_VO keeps the per-frame mirror path)."""
    from ..pipeline import RGBDCamConfig, RGBDSequenceEngine
    if (tracker.detection_method != "GFT" or tracker.matching_type != "BF" or tracker.k_best_matches != 1
            or tracker.use_descriptor_radius_match_for_motion):
        return None
    cm = camera_model
    cam = RGBDCamConfig(cm.fx, cm.fy, cm.center_x, cm.center_y, cm.focal_length_m, cm.depth_is_Z, 0.8, 7.0,
                        f2f_max_hdiff=tracker.max_horizontal_diff_f2f_matches, pct_good_matches=tracker.percentage_good_matches)
    shape = tuple(int(v) for v in np.asarray(first_rgb).shape[:2])
    # (as sequence_engine_for: the engine -- frame store, 136 MB of pinned staging buffers, scratch -- stays with the camera
    # model and serves later runs with the same settings; building one costs 15 ms, a 128-frame run 45)
    key = (int(window), shape, int(tracker.num_features_detection_for_motion), float(tracker.backprojection_score_threshold_3D_to_2D),
           int(tracker.max_ransac_iterations_3D_to_2D), float(tracker.max_horizontal_diff_f2f_matches),
           float(tracker.percentage_good_matches), str(tracker.pose_est_algorithm), float(cm.fx), float(cm.fy), float(cm.center_x),
           float(cm.center_y), float(cm.focal_length_m), bool(cm.depth_is_Z))
    cache = cm.__dict__.setdefault("_sequence_engines", {})
    eng = cache.get(key)
    if eng is None:
        cache.clear()
        eng = cache[key] = RGBDSequenceEngine(cm._context(), cam, window=window, image_shape=shape,
                                              num_of_features=tracker.num_features_detection_for_motion, median_win_size=0,
                                              thr=tracker.backprojection_score_threshold_3D_to_2D,
                                              max_iter=tracker.max_ransac_iterations_3D_to_2D, adaptive=True,
                                              lm_iter=pyopengv.LM_MAX_ITERATIONS, pose_est_algorithm=tracker.pose_est_algorithm)
    else:
        eng.reset()
    return eng


class TrackerRGBDSE3(TrackerSE3):
    """pose_est_tools.py:880-958: central 3D-2D tracking of RGB-D frames."""

    def __init__(self, camera_model, show_3D_points=False, **kwargs):
        TrackerSE3.__init__(self, camera_model, show_3D_points, **kwargs)
        self.number_of_cams = 1
        self.T_C_wrt_S_init = np.identity(4)
        self.T_C_curr_frame_wrt_S_est = kwargs.get("T_C_wrt_S_init", self.T_C_wrt_S_init)
        self.bootstrap_tracker()

    def bootstrap_tracker(self):
        """pose_est_tools.py:956-958"""
        self.camera_model.feature_matcher_for_motion = FeatureMatcher(
            method=self.detection_method, matcher_type=self.matching_type, k_best=self.k_best_matches,
            percentage_good_matches=self.percentage_good_matches, num_of_features=self.num_features_detection_for_motion,
            use_radius_match=self.use_descriptor_radius_match_for_motion)
        self.max_horizontal_diff_f2f_matches = self.max_horizontal_search_ratio * (self.camera_model.center_x * 2.)

    def track_frame(self, reference_frame, current_frame):
        """pose_est_tools.py:896-954 -> (ok, message)."""
        self.num_tracked_correspondences = 0
        self.inlier_tracked_correspondences_ratio = 0.
        if isinstance(current_frame, DeviceRGBDFrame):
            return self._track_device_frame(reference_frame, current_frame)
        ref, cur = reference_frame.keypoints_and_descriptors, current_frame.keypoints_and_descriptors
        (t_idx, _, _), (q_idx, _, _), _ = match_features_frame_to_frame(
            cam_model=self.camera_model, train_kpts=ref.keypoints, train_desc=ref.descriptors, query_kpts=cur.keypoints,
            query_desc=cur.descriptors, random_colors_RGB=ref.random_colors_RGB,
            keypts_as_points_train=ref.pixel_coords, keypts_as_points_query=cur.pixel_coords,
            max_horizontal_diff=self.max_horizontal_diff_f2f_matches, max_descriptor_distance_radius=-1)
        t_idx, q_idx = np.asarray(t_idx, dtype=np.int64), np.asarray(q_idx, dtype=np.int64)
        num_initial_matches = len(t_idx)
        if num_initial_matches < 2 * self.n_points_for_RANSAC_model * self.number_of_cams:
            return False, "Cannot track on only %d point correspondences" % (num_initial_matches)
        b = current_frame.bearing_vectors[q_idx]
        p = reference_frame.keypoints_3D_points[t_idx]
        T_ransac, inliers = pyopengv.absolute_pose_ransac(b[..., :3], p[..., :3], self.pose_est_algorithm,
                                                          self.backprojection_score_threshold_3D_to_2D,
                                                          self.max_ransac_iterations_3D_to_2D)
        self.num_tracked_correspondences = len(inliers)
        self.inlier_tracked_correspondences_ratio = float(self.num_tracked_correspondences) / float(num_initial_matches)
        if self.num_tracked_correspondences < self.n_points_for_RANSAC_model:
            # no consensus model (e.g. EPNP on an exactly coplanar point set, where it has no solution): report the
            # frame as not tracked instead of refining from an empty inlier set
            return False, "RANSAC (%s) found no model among %d correspondences" % (self.pose_est_algorithm, num_initial_matches)
        T_nl = pyopengv.absolute_pose_optimize_nonlinear(b[inliers], p[inliers], T_ransac[:3, 3], T_ransac[:3, :3])
        T_homo = np.identity(4)
        T_homo[:3] = T_nl
        T_homo[:3, 3] = T_nl[:3, 3] * current_frame.conversion_factor_length_to_m
        current_frame.T_frame_wrt_tracking_ref_frame = T_homo
        T_key = self.T_Ckey_wrt_S_est_list[-1] if self.T_Ckey_wrt_S_est_list else np.identity(4)
        self.T_C_curr_frame_wrt_S_est = tr.concatenate_matrices(T_key, T_homo)
        if self.show_3D_points:
            self.xyz_homo_points_wrt_C_inliers = np.hstack((p[inliers] * current_frame.conversion_factor_length_to_m,
                                                            np.ones((len(inliers), 1))))
        return True, "tracking used %d inlier point correspondences" % (self.num_tracked_correspondences)


def _rgbd_track_device_frame(self, reference_frame, current_frame):
    """TrackerRGBDSE3.track_frame on frames of an RGBDSequenceEngine's store: the [16] record of the pair (the speculative
    one when the reference is the frame's predecessor, one serial call against the keyframe slot otherwise) put through the
    bookkeeping of track_frame above."""
    eng = current_frame.engine
    ref_seq = getattr(reference_frame, "seq_index", None)
    if current_frame.spec_record is not None and ref_seq == current_frame.seq_index - 1:
        rec = current_frame.spec_record
    elif current_frame.spec2_record is not None and ref_seq == current_frame.seq_index - 2:
        rec = current_frame.spec2_record
    else:
        rec = eng.track(eng.key_slot, current_frame.slot, current_frame.seed)
    num_initial_matches = int(rec[13])
    if num_initial_matches < 2 * self.n_points_for_RANSAC_model * self.number_of_cams:
        return False, "Cannot track on only %d point correspondences" % (num_initial_matches)
    self.num_tracked_correspondences = int(rec[12])
    self.inlier_tracked_correspondences_ratio = float(self.num_tracked_correspondences) / float(num_initial_matches)
    if self.num_tracked_correspondences < self.n_points_for_RANSAC_model:
        return False, "RANSAC (%s) found no model among %d correspondences" % (self.pose_est_algorithm, num_initial_matches)
    T_homo = np.identity(4)
    T_homo[:3] = rec[:12].reshape(3, 4)
    T_homo[:3, 3] = T_homo[:3, 3] * current_frame.conversion_factor_length_to_m
    current_frame.T_frame_wrt_tracking_ref_frame = T_homo
    T_key = self.T_Ckey_wrt_S_est_list[-1] if self.T_Ckey_wrt_S_est_list else np.identity(4)
    self.T_C_curr_frame_wrt_S_est = tr.concatenate_matrices(T_key, T_homo)
    return True, "tracking used %d inlier point correspondences" % (self.num_tracked_correspondences)


TrackerRGBDSE3._track_device_frame = _rgbd_track_device_frame


def _is_rgbd_model(camera_model):
    from .camera_models import RGBDCamModel
    return isinstance(camera_model, RGBDCamModel)


def run_VO(visualizer_3D_VO, camera_model, gt_poses_filename=None, est_poses_filename="estimated_frame_poses_TUM.txt",
           img_filename_template=None, depth_filename_template=None, img_indices=(), results_path="~/temp", thread_name="",
           _live_frames=None, _keyframe_thresholds=None, frame_window=None):
    """pose_est_tools.py:1264-1678 without the 3-D visualisation (visualizer_3D_VO must be None): the frame loop,
    the keyframe policy (translation 0.01-0.20 m or rotation 1-10 degrees wrt the keyframe, enough tracked
    correspondences and keypoints), pose chaining through the keyframes, and the result files
        estimated_frame_poses_TUM.txt, gt_associated_frame_poses_TUM.txt  "idx tx ty tz qx qy qz qw" [m]
        keyframe_ids.txt, printed_messages.log
    -> dict(poses=[(idx, T 4x4)], keyframe_ids=[...], tracked=number of tracked frames, message=summary).

    frame_window: SEQUENCE MODE (pipeline.SequenceEngine / RGBDSequenceEngine): the front ends of `frame_window` frames
    run as ONE batch on the GPU and the frames are tracked from the device-resident frame store (speculatively against
    their predecessors, serially against the keyframe where that guess fails).  None = 32 for an image sequence, 1 for a
    live source; 0 = the per-frame mirror path (StereoPanoramicFrame on host arrays, one set of C-ABI calls per stage).
    The pose file does not depend on the window size (byte-identical for 1, 2, 32, ...); the mirror path agrees with
    it to rounding (its bearings go through numpy's trigonometry, the store's through the library's)."""
    from .common_cv import get_depthmap_float32_from_png, get_images, imread
    from .common_tools import get_poses_from_file
    if visualizer_3D_VO is not None:
        raise NotImplementedError("3-D visualisation is not built: pass visualizer_3D_VO=None")
    pyopengv.set_seed(0)  # a run is a pure function of its inputs (OpenGV seeds its sampler from the clock)
    prefix = thread_name + ": " if len(thread_name) > 0 else ""
    rgbd = _is_rgbd_model(camera_model)
    trackerClass, KeyFrameClass = (TrackerRGBDSE3, RGBDKeyFrame) if rgbd else (TrackerStereoSE3, StereoPanoramicKeyFrame)
    results_path = os.path.realpath(os.path.expanduser(results_path))
    make_sure_path_exists(results_path)
    log = open(os.path.join(results_path, "printed_messages.log"), "w")
    # Sequence mode keeps the per-frame host work small: messages and pose lines are collected and written once per window
    # (same text, same order), the quaternions of a window's poses come from ONE batched eigen-decomposition
    # (tr.quaternions_from_matrices: bit-identical rows).  The per-frame paths (frame_window 0, live sources) flush every frame.
    out_msgs, log_only, pending = [], [], []    # stdout + log lines | (position in out_msgs, line) for the log only | poses

    def say(msg):
        out_msgs.append(msg)

    def flush_output():
        if out_msgs or log_only:
            if out_msgs:
                print("\n".join(out_msgs))
            lines, extra = list(out_msgs), sorted(log_only, key=lambda e: e[0], reverse=True)
            for pos, line in extra:            # (a failure message goes to the log only, where it was said)
                lines.insert(pos, line)
            log.write("\n".join(lines) + "\n")
            del out_msgs[:], log_only[:]
        if pending:
            idxs = [p_[0] for p_ in pending]
            Te = np.stack([p_[1] for p_ in pending])
            qe, te = tr.quaternions_from_matrices(Te).tolist(), Te[:, :3, 3].tolist()
            est_file.write("".join("%d %s %s %s %s %s %s %s\n" % (i, repr(t[0]), repr(t[1]), repr(t[2]), repr(q[1]), repr(q[2]),
                                                                     repr(q[3]), repr(q[0])) for i, t, q in zip(idxs, te, qe)))
            gts = [p_[2] for p_ in pending]
            uniq = {}
            for g in gts:                        # (one shared identity for a run without ground truth: decomposed once)
                uniq.setdefault(id(g), g)
            keys = list(uniq)
            Tg = np.stack([uniq[k_] for k_ in keys])
            bad = np.isnan(Tg).any(axis=(1, 2))
            qg = tr.quaternions_from_matrices(np.where(bad[:, None, None], np.identity(4), Tg)).tolist()
            tg = Tg[:, :3, 3].tolist()
            line_of = {}
            for k_, b_, t, q in zip(keys, bad.tolist(), tg, qg):
                line_of[k_] = " nan nan nan nan nan nan nan\n" if b_ else " %s %s %s %s %s %s %s\n" % (
                    repr(t[0]), repr(t[1]), repr(t[2]), repr(q[1]), repr(q[2]), repr(q[3]), repr(q[0]))
            gt_file.write("".join("%d%s" % (i, line_of[id(g)]) for i, g in zip(idxs, gts)))
            del pending[:]

    pose_output_file_units = "m"
    zero_up_gt_wrt_origin = True
    # indoor thresholds (pose_est_tools.py:1308-1313); run_VO_live passes its own (:991-1003)
    pos_thr, pos_max, ang_thr, ang_max = _keyframe_thresholds or (0.01, 0.20, np.deg2rad(1.0), np.deg2rad(10.0))
    thr_tracked, thr_keypoints = 0.10, 0.10

    if _live_frames is None:
        image_names = get_images(img_filename_template, indices_list=img_indices, return_names_only=True)
        depth_names = get_images(depth_filename_template, indices_list=img_indices, return_names_only=True) if rgbd else None
        if img_indices is None or len(img_indices) == 0:
            img_indices = list(range(len(image_names)))

        def load(k):
            if rgbd:
                # the reference converts BGR -> RGB (:1430) and then treats the array as BGR (:531): reproduced as is
                return img_indices[k], np.ascontiguousarray(imread(image_names[k])[..., ::-1]), \
                    get_depthmap_float32_from_png(depth_names[k], camera_model.scaling_factor)
            return img_indices[k], imread(image_names[k]), None

        def frames():   # (frame index, image, depth map or None) from the files of the sequence; the next frame is
            # decoded on a helper thread while the current one is tracked (the decoder releases the GIL)
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=1) as pool:
                pending = pool.submit(load, 0) if len(img_indices) else None
                for k in range(len(img_indices)):
                    item = pending.result()
                    pending = pool.submit(load, k + 1) if k + 1 < len(img_indices) else None
                    yield item
    else:
        img_indices = [0]
        frames = _live_frames
    tracker = trackerClass(camera_model=camera_model, show_3D_points=False, results_path=results_path)
    if frame_window is None:
        frame_window = 32 if _live_frames is None else 1
    engine_state = dict(engine=None, tried=False, window_s=0.0, windows=0)
    if int(frame_window) > 0:
        source = frames

        def frames():   # the same frames, `frame_window` at a time through the batched front end
            import queue
            W = int(frame_window)
            it = iter(source())

            def next_chunk():
                chunk = []
                for item in it:
                    chunk.append(item)
                    if len(chunk) >= W:
                        break
                return chunk
            chunk = next_chunk()
            if not chunk:
                return
            engine_state["tried"] = True
            eng = engine_state["engine"] = (rgbd_sequence_engine_for if rgbd else sequence_engine_for)(
                tracker, camera_model, chunk[0][1], W)
            if eng is None:              # no batched counterpart of this configuration: the per-frame path
                while chunk:
                    for item in chunk:
                        yield item
                    chunk = next_chunk()
                return
            images_of = (lambda c: [(item[1], item[2]) for item in c]) if rgbd else (lambda c: [item[1] for item in c])
            # a helper thread pulls the NEXT window from the source and copies it into the engine's other pinned buffer while
            # this one is processed (the copy releases the GIL); a buffer returns to the thread after its upload was synchronised
            free, ready = queue.Queue(), queue.Queue()
            free.put(1)
            stop = threading.Event()

            def producer():
                try:
                    while not stop.is_set():
                        c = next_chunk()
                        if not c:
                            break
                        b = free.get()
                        if b is None:
                            return
                        t0 = time.perf_counter()
                        eng.stage_host(images_of(c), b)
                        eng.stage_s["stage_to_pinned"] += time.perf_counter() - t0
                        if eng.early_upload:
                            eng.upload_staged(b, len(c))   # the copy to the device starts now, beside the current window's kernels
                        ready.put((c, b))
                    ready.put((None, None))
                except BaseException as e:   # handed to the consumer
                    ready.put((e, None))
            t0 = time.perf_counter()
            eng.stage_host(images_of(chunk), 0)
            eng.stage_s["stage_to_pinned"] += time.perf_counter() - t0
            worker = threading.Thread(target=producer, name="sosvo-stage", daemon=True)
            worker.start()
            seq_index, buf = 0, 0
            try:
                t0 = time.perf_counter()
                eng.enqueue_staged(buf, len(chunk))
                engine_state["window_s"] += time.perf_counter() - t0
                while chunk:
                    t0 = time.perf_counter()
                    infos = eng.collect()                        # (synchronises: the upload from `buf` is complete)
                    engine_state["windows"] += 1
                    free.put(buf)
                    # the next window goes onto the GPU BEFORE this one's records are worked through, when the helper thread
                    # has it staged already (an image sequence: nearly always; a live source: never -- its frames are not
                    # held back for it)
                    nxt = None
                    if eng.enqueue_ahead:
                        try:
                            nxt = ready.get_nowait()
                        except queue.Empty:
                            nxt = None
                        if nxt is not None and nxt[0] is not None and not isinstance(nxt[0], BaseException):
                            eng.enqueue_staged(nxt[1], len(nxt[0]))
                    engine_state["window_s"] += time.perf_counter() - t0
                    flush_output()                               # the previous window's lines, off the per-frame path
                    for item, info in zip(chunk, infos):
                        yield item[0], item[1], None, info, seq_index
                        seq_index += 1
                    if nxt is None:
                        nxt = ready.get()
                        if nxt[0] is not None and not isinstance(nxt[0], BaseException):
                            t0 = time.perf_counter()
                            eng.enqueue_staged(nxt[1], len(nxt[0]))
                            engine_state["window_s"] += time.perf_counter() - t0
                    chunk, buf = nxt
                    if isinstance(chunk, BaseException):
                        raise chunk
            finally:
                stop.set()
                free.put(None)
    if gt_poses_filename is None or not os.path.exists(gt_poses_filename):
        n_gt = max(len(img_indices), img_indices[-1] + 1)
        gt_list = n_gt * [np.identity(4)]   # (a live run has no ground truth: identity for every frame, as :1012)
    else:
        _, gt_list = get_poses_from_file(poses_filename=gt_poses_filename, input_units="m",
                                         output_working_units=pose_output_file_units, indices=[], pose_format="tum",
                                         zero_up_wrt_origin=zero_up_gt_wrt_origin)
    T_Rgt_wrt_S = None
    if gt_poses_filename is not None and os.path.exists(gt_poses_filename) and getattr(camera_model, "T_Cest_wrt_Rgt", None) is not None:
        # fixed hand-eye transformation between the estimated camera frame and the ground-truth rig frame (:1362-1367)
        k_units = get_length_units_conversion_factor(camera_model.units, pose_output_file_units)
        T_Cest_wrt_Rgt = np.array(camera_model.T_Cest_wrt_Rgt, dtype=np.float64, copy=True)
        T_Cest_wrt_Rgt[:3, 3] = k_units * T_Cest_wrt_Rgt[:3, 3]
        T_Rgt_wrt_S = tr.concatenate_matrices(tracker.T_C_wrt_S_init, tr.inverse_matrix(T_Cest_wrt_Rgt))
        T_S_wrt_Rgt = tr.inverse_matrix(T_Rgt_wrt_S)
    est_path = os.path.join(results_path, est_poses_filename)
    est_file = open(est_path, "w")
    gt_file = open(est_path.replace("estimated", "gt_associated"), "w")
    kf_path = os.path.join(results_path, "keyframe_ids.txt")
    kf_file = open(kf_path, "w")

    reference_frame, current_frame = None, None
    create_keyframe = True
    current_keyframe_id = img_indices[0]
    number_of_keyframes = 0
    tracked_wrt_keyframe = 0
    tracked_prev_avg = 0.
    inlier_ratio_cma = tracker.inlier_tracked_correspondences_ratio
    acc = dict(read=0., setup=0., track=0., vo=0., corr=0., ratio=0.)
    poses_out, keyframe_ids = [], []
    n_done = 0
    frame_iter = iter(frames())
    img_index_number = -1
    batched_output = False       # sequence mode: lines are written once per window (flush_output in the window generator)
    kf_lines = []
    while True:
        t_frame = t0 = time.process_time()
        try:
            item = next(frame_iter)
        except StopIteration:
            break
        idx, img, depth_map = item[:3]
        dev_info = item[3] if len(item) > 3 else None
        batched_output = dev_info is not None
        img_index_number += 1
        t1 = time.process_time()
        if img_index_number > 0:
            acc["read"] += t1 - t0
        t0 = t1
        if dev_info is not None:
            current_frame = (DeviceRGBDFrame if rgbd else DeviceStereoFrame)(
                engine_state["engine"], dev_info, camera_model, frame_id=idx, seq_index=item[4], parent_id=current_keyframe_id)
        elif rgbd:
            current_frame = RGBDFrame(rgbd_camera_model=camera_model, frame_id=idx, rgb_img=img, depth_map=depth_map,
                                      parent_id=current_keyframe_id)
        else:
            camera_model.set_current_omni_image(img, generate_panoramas=False, view=False, apply_mask=True, mask_RGB=(0, 0, 0))
            current_frame = StereoPanoramicFrame(stereo_camera_model=camera_model, frame_id=idx, parent_id=current_keyframe_id)
        t1 = time.process_time()
        if img_index_number > 0:
            acc["setup"] += t1 - t0
        T_gt = gt_list[idx] if idx < len(gt_list) else (np.identity(4) if _live_frames is not None else np.full((4, 4), np.nan))
        if T_Rgt_wrt_S is not None:
            T_gt = tr.concatenate_matrices(T_Rgt_wrt_S, T_gt, T_S_wrt_Rgt)   # :1472
        if img_index_number > 0:
            t0 = t1
            ok, msg = tracker.track_frame(reference_frame=reference_frame, current_frame=current_frame)
            if not ok:
                log_only.append((len(out_msgs), msg))
                warn("%sWarning failed: %s" % (prefix, msg))
                break
            reference_frame.children_ids.append(current_frame.frame_id)
            tracked_wrt_keyframe += 1
            acc["track"] += time.process_time() - t0
            # keyframe policy (:1516-1550)
            dist = tr.rpe_translation_metric(current_frame.T_frame_wrt_tracking_ref_frame)
            ang = tr.rpe_rotation_metric(current_frame.T_frame_wrt_tracking_ref_frame)
            n_tracked = tracker.num_tracked_correspondences / float(tracker.number_of_cams)
            acc["corr"] += n_tracked
            M_K, M_F = reference_frame.num_valid_keypoints, current_frame.num_valid_keypoints
            if (pos_thr < dist < pos_max) or (ang_thr < ang < ang_max):
                if n_tracked > thr_tracked * tracked_prev_avg and M_F > thr_keypoints * M_K:
                    reason = ""
                    if pos_thr < dist < pos_max:
                        if ang < ang_max:
                            reason += "translation %.2f < %.4f < %.2f [m]" % (pos_thr, dist, pos_max)
                            create_keyframe = True
                    elif ang_thr < ang:
                        if dist < pos_max < ang_max:   # as written in the reference (:1534)
                            reason += " rotation: %.2f > %.2f [radians]" % (ang, ang_thr)
                            create_keyframe = True
                    if create_keyframe:
                        say("%sFrame [%d] as Keyframe due to %s... with %d tracked keypoint correspondences after tracking "
                            "%d frames" % (prefix, idx, reason, n_tracked, tracked_wrt_keyframe))
                    else:
                        say("%sFrame [%d] as failed to create Keyframe due to some CRAZYNESS" % (prefix, idx))
                else:
                    say("%sFrame [%d] as Keyframe...doesn't satisfy %d > %.2f * %.2f and %d > %.2f * %.2f"
                        % (prefix, idx, n_tracked, thr_tracked, tracked_prev_avg, M_F, thr_keypoints, M_K))
            tracked_prev_avg = (n_tracked + (float(tracked_wrt_keyframe) - 1.) * tracked_prev_avg) / float(tracked_wrt_keyframe)
            acc["ratio"] += tracker.inlier_tracked_correspondences_ratio
            inlier_ratio_cma = (tracker.inlier_tracked_correspondences_ratio + float(img_index_number - 1) * inlier_ratio_cma) \
                / float(img_index_number)
        if create_keyframe:
            tracked_wrt_keyframe = 0
            tracked_prev_avg = 0.
            reference_frame = KeyFrameClass(frame=current_frame)
            if dev_info is not None:
                current_frame.promote()   # the frame's record moves to the store's keyframe slot
            current_keyframe_id = reference_frame.frame_id
            kf_lines.append("%d\n" % current_keyframe_id)
            keyframe_ids.append(current_keyframe_id)
            if len(tracker.T_Ckey_wrt_S_est_list) > 0:
                T_key = tr.concatenate_matrices(tracker.T_Ckey_wrt_S_est_list[-1], reference_frame.T_frame_wrt_tracking_ref_frame)
            else:
                T_key = tracker.T_C_curr_frame_wrt_S_est
            tracker.T_Ckey_wrt_S_est_list.append(T_key)
            create_keyframe = False
            number_of_keyframes += 1
        # TUM lines: idx tx ty tz qx qy qz qw (:1611-1621), quaternion_from_matrix(isprecise=False) of the pose -- formatted by
        # flush_output (a window at a time in sequence mode)
        T_now = np.array(tracker.T_C_curr_frame_wrt_S_est, copy=True)
        pending.append((idx, T_now, T_gt))
        poses_out.append((idx, T_now))
        if img_index_number > 0:
            acc["vo"] += time.process_time() - t_frame
        n_done = img_index_number
        say("%sDONE with F[%d] (Parent K[%d]). C.M.Avg. RANSAC inlier ratio = %.3f"
            % (prefix, current_frame.frame_id, current_frame.parent_id, inlier_ratio_cma))
        if not batched_output:
            flush_output()
    flush_output()
    kf_file.write("".join(kf_lines))
    d = float(max(n_done, 1))
    summary = "\n".join([
        "%sVO done with %d keyframes" % (prefix, number_of_keyframes),
        "Image Read Avg Time: {time:.8f} seconds".format(time=acc["read"] / d),
        "Frame Setup Avg Time: {time:.8f} seconds".format(time=acc["setup"] / d),
        "Frame Tracking Avg Time: {time:.8f} seconds".format(time=acc["track"] / d),
        "Overall Frame VO Avg Time: {time:.8f} seconds".format(time=acc["vo"] / d),
        "Total Number of tracked correspondences: {corrs}".format(corrs=int(acc["corr"])),
        "Average Number of tracked correspondences: {corrs}".format(corrs=int(int(acc["corr"]) / d)),
        "Average tracked correspondences RANSAC inlier ratio: {ratio:.8f}".format(ratio=acc["ratio"] / d),
        "Estimated poses in TUM format SAVED as " + est_path,
        "Keyframes (indices) SAVED as " + kf_path])
    say(summary)
    flush_output()
    for f in (est_file, gt_file, kf_file, log):
        f.close()
    out = dict(poses=poses_out, keyframe_ids=keyframe_ids, tracked=n_done, message=summary)
    if engine_state["engine"] is not None:
        eng = engine_state["engine"]
        out["sequence_mode"] = dict(frame_window=eng.W, windows=engine_state["windows"], serial_tracking_calls=eng.serial_calls,
                                    front_end_and_speculation_s=engine_state["window_s"], stage_s=dict(eng.stage_s))
    return out


def run_VO_live(visualizer_3D_VO, camera_model, cam_working_thread, est_poses_filename="estimated_frame_poses_TUM.txt",
                results_path="~/temp", thread_name="", frame_window=None):
    """pose_est_tools.py:960-1262: the VO loop on the frames of a running camera thread.  `cam_working_thread` is the
    reference's CamAsWorkingThread contract (webcam_live.py:256-291; `omnistereo.webcam_live.FrameSourceThread` here):
    `.current_frame` = the most recent omni image (None when the source ends), `.quit_flag`.  Each pass of the loop
    takes whatever frame is current (frames arriving faster than the tracker are skipped, a frame still current is
    processed again, as in the reference), numbers it 0, 1, 2, ... and runs run_VO's frame body with the live keyframe
    thresholds (:998-1003: translation 0.05-0.50 m or rotation 5-60 degrees); omnistereo models only (:1066-1073)."""
    if _is_rgbd_model(camera_model):
        raise NotImplementedError("run_VO_live tracks omnistereo models only (the RGB-D branch of the reference is commented out, :1069-1073)")

    def frames():
        idx = 0
        while not cam_working_thread.quit_flag:   # :1051
            img = cam_working_thread.current_frame
            if img is None:
                break
            yield idx, img, None
            idx += 1
    return run_VO(visualizer_3D_VO, camera_model, gt_poses_filename=None, est_poses_filename=est_poses_filename,
                  results_path=results_path, thread_name=thread_name, _live_frames=frames,
                  _keyframe_thresholds=(0.05, 0.50, np.deg2rad(5.0), np.deg2rad(60.0)), frame_window=frame_window)


def driver_VO_live(camera_model, scene_path_vo_results, cam_working_thread, visualize_VO=False, use_multithreads_for_VO=True,
                   thread_name="LIVE", frame_window=None):
    """pose_est_tools.py:1743-1797: start the camera thread, run run_VO_live (on a worker thread or inline), then stop
    and join the camera thread."""
    if visualize_VO:
        raise NotImplementedError("3-D visualisation is not built: visualize_VO must be False")
    if use_multithreads_for_VO:
        est_poses_filename = "estimated_frame_poses_TUM.txt"
    else:
        from datetime import datetime
        now = datetime.now()
        est_poses_filename = "estimated_frame_poses_TUM-%d-%d-%d-%d-%d-%d.txt" % (now.year, now.month, now.day, now.hour,
                                                                                   now.minute, now.second)
    cam_working_thread.start()
    if not getattr(cam_working_thread, "lockstep", False):
        while cam_working_thread.current_frame is None and cam_working_thread.is_alive() and not cam_working_thread.quit_flag:
            time.sleep(0.001)   # (the reference starts tracking at once and stops if no frame has arrived yet)
    kwargs = dict(visualizer_3D_VO=None, camera_model=camera_model, cam_working_thread=cam_working_thread,
                  est_poses_filename=est_poses_filename, results_path=scene_path_vo_results, thread_name=thread_name,
                  frame_window=frame_window)
    result, failure = {}, []
    try:
        if use_multithreads_for_VO:
            def work():
                try:
                    result.update(run_VO_live(**kwargs))
                except BaseException as e:  # noqa: B902 -- handed to the calling thread
                    failure.append(e)
            th = threading.Thread(target=work)
            th.start()
            th.join()
        else:
            result.update(run_VO_live(**kwargs))
    finally:
        cam_working_thread.quit_flag = True   # :1792-1793
        cam_working_thread.join()
    if failure:
        raise failure[0]
    print("DONE")
    return result


def driver_VO(camera_model, scene_path, scene_path_vo_results, scene_img_filename_template, depth_filename_template,
              num_scene_images, visualize_VO=False, use_multithreads_for_VO=True, step_for_scene_images=1, first_image_index=0,
              last_image_index=-1, thread_name="", frame_window=None):
    """pose_est_tools.py:1680-1741: frame index list, ground truth file discovery (gt_TUM.txt unless the scene is a
    static one), run_VO on a worker thread (as the reference does next to its GUI loop) or inline."""
    if visualize_VO:
        raise NotImplementedError("3-D visualisation is not built: visualize_VO must be False")
    if use_multithreads_for_VO:
        est_poses_filename = "estimated_frame_poses_TUM.txt"
    else:
        from datetime import datetime
        now = datetime.now()
        est_poses_filename = "estimated_frame_poses_TUM-%d-%d-%d-%d-%d-%d.txt" % (now.year, now.month, now.day, now.hour,
                                                                                   now.minute, now.second)
    gt_poses_filename = None
    low = scene_path.lower()
    rig_is_static = "static" in low or "GCT" in scene_path.upper() or "CCNY" in scene_path.upper() or "park" in low \
        or "grand" in low
    if not rig_is_static:
        gt_poses_filename = os.path.join(scene_path, "gt_TUM.txt")
    last_image_index = min(last_image_index, num_scene_images) if last_image_index > 0 else num_scene_images
    vo_frame_indices = list(range(first_image_index, last_image_index, step_for_scene_images))
    kwargs = dict(visualizer_3D_VO=None, camera_model=camera_model, gt_poses_filename=gt_poses_filename,
                  est_poses_filename=est_poses_filename, img_filename_template=scene_img_filename_template,
                  depth_filename_template=depth_filename_template, img_indices=vo_frame_indices,
                  results_path=scene_path_vo_results, thread_name=thread_name, frame_window=frame_window)
    result = {}
    if use_multithreads_for_VO:
        # one libsosvo context per host thread: the worker creates its own
        failure = []

        def work():
            try:
                result.update(run_VO(**kwargs))
            except BaseException as e:  # noqa: B902 -- handed to the calling thread
                failure.append(e)
        th = threading.Thread(target=work)
        th.start()
        th.join()
        if failure:
            raise failure[0]
    else:
        result.update(run_VO(**kwargs))
    print("%s Done with VO for %s!" % (thread_name, scene_path))
    return result
