"""Host-side mirror of the hot-path types of omnistereo/camera_models.py: same names, arguments and return
structure, the arithmetic behind them on the GPU through libsosvo's C ABI.

    KeyPointAndDescriptor      camera_models.py:250-289
    PanoramicCorrespondences   camera_models.py:291-362
    FeatureMatcher             camera_models.py:364-446  (cv2.BFMatcher(NORM_HAMMING) -> sosvo_match_hamming +
                                                          sosvo_sort_matches)

cv2 is not a dependency here, so the two cv2 value types the reference passes around are small Python classes
with the same attribute names: KeyPoint (.pt .size .angle .response .octave .class_id) and DMatch (.queryIdx
.trainIdx .imgIdx .distance).  There is no CPU fallback: FeatureMatcher.match needs the GPU library."""
import numpy as np


class KeyPoint(object):
    """cv2.KeyPoint's attribute set."""
    __slots__ = ("pt", "size", "angle", "response", "octave", "class_id")

    def __init__(self, x, y, size=31.0, angle=-1.0, response=0.0, octave=0, class_id=-1):
        self.pt = (float(x), float(y))
        self.size, self.angle, self.response = float(size), float(angle), float(response)
        self.octave, self.class_id = int(octave), int(class_id)

    def __repr__(self):
        return "KeyPoint(%.1f, %.1f)" % self.pt


class DMatch(object):
    """cv2.DMatch's attribute set."""
    __slots__ = ("queryIdx", "trainIdx", "imgIdx", "distance")

    def __init__(self, queryIdx, trainIdx, distance, imgIdx=0):
        self.queryIdx, self.trainIdx, self.imgIdx = int(queryIdx), int(trainIdx), int(imgIdx)
        self.distance = float(distance)

    def __repr__(self):
        return "DMatch(q=%d, t=%d, d=%g)" % (self.queryIdx, self.trainIdx, self.distance)


class MatchList(list):
    """What FeatureMatcher.match returns: a list of DMatch (so reference-style loops work) that also carries
    the same information as arrays (.query_idx, .train_idx, .distances) for vectorised callers."""

    def __init__(self, query_idx, train_idx, distances):
        self.query_idx = np.asarray(query_idx, dtype=np.int64)
        self.train_idx = np.asarray(train_idx, dtype=np.int64)
        self.distances = np.asarray(distances, dtype=np.float32)
        list.__init__(self, [DMatch(q, t, d) for q, t, d in zip(self.query_idx, self.train_idx, self.distances)])


def keypoints_to_array(kpts):
    """cv2.KeyPoint_convert(kpts): [n,2] float32 of .pt."""
    if len(kpts) == 0:
        return np.empty((0, 2), dtype=np.float32)
    return np.array([k.pt for k in kpts], dtype=np.float32)


def _flatten(list_of_lists):
    out = []
    for item in list_of_lists:
        if item is None:
            continue
        out.extend(item)
    return out


class KeyPointAndDescriptor(object):
    """camera_models.py:250-289."""

    def __init__(self, kpts_list, desc_list, coords_array=None, random_colors_RGB_list=[], do_flattening=False, **kwargs):
        if do_flattening:
            self.keypoints = _flatten(kpts_list)
            self.descriptors = np.array(_flatten(desc_list))
        else:
            self.keypoints = kpts_list
            self.descriptors = desc_list
        n = len(self.keypoints)
        if coords_array is None or len(coords_array) == 0:
            self.pixel_coords = np.ones((1, n, 3))
            if n:
                self.pixel_coords[0, :, :2] = keypoints_to_array(self.keypoints)
        else:
            self.pixel_coords = coords_array
        if random_colors_RGB_list is None or len(random_colors_RGB_list) < n:
            self.random_colors_RGB = np.random.randint(low=0, high=256, size=(n, 3), dtype="uint8")
        else:
            self.random_colors_RGB = random_colors_RGB_list


class PanoramicCorrespondences(object):
    """camera_models.py:291-362: the per-frame container between frame set-up and tracking."""

    def __init__(self, kpts_top_list, desc_top_list, kpts_bot_list, desc_bot_list, points_3D=None, m_top_array=None,
                 m_bot_array=None, random_colors_RGB_list=[], do_flattening=False, **kwargs):
        if do_flattening:
            self.kpts_top, self.kpts_bot = _flatten(kpts_top_list), _flatten(kpts_bot_list)
            self.desc_top, self.desc_bot = np.array(_flatten(desc_top_list)), np.array(_flatten(desc_bot_list))
        else:
            self.kpts_top, self.kpts_bot = kpts_top_list, kpts_bot_list
            self.desc_top, self.desc_bot = desc_top_list, desc_bot_list
        if points_3D is not None:
            if len(points_3D) > 0:
                if points_3D.shape[-1] == 3:
                    self.points_3D_coords_homo = np.ones((len(points_3D), 4))
                    self.points_3D_coords_homo[:, :3] = points_3D[:, :3]
                else:
                    self.points_3D_coords_homo = points_3D
            else:
                self.points_3D_coords_homo = np.empty((0, 4))
        else:
            self.points_3D_coords_homo = []
        self.m_top = self._homogeneous_pixels(self.kpts_top, m_top_array)
        self.m_bot = self._homogeneous_pixels(self.kpts_bot, m_bot_array)
        n = max(len(self.kpts_top), len(self.kpts_bot))
        if random_colors_RGB_list is None or len(random_colors_RGB_list) < n:
            self.random_colors_RGB = np.random.randint(low=0, high=256, size=(n, 3), dtype="uint8")
        else:
            self.random_colors_RGB = random_colors_RGB_list

    @staticmethod
    def _homogeneous_pixels(kpts, given):
        if given is not None and len(given) > 0:
            return given
        if len(kpts) > 0:
            m = keypoints_to_array(kpts).astype(np.float64)
            return np.hstack((m, np.ones_like(m[..., 0, np.newaxis])))
        return np.empty((0, 3))


class FeatureMatcher(object):
    """camera_models.py:364-446.  Brute-force matching on the GPU: Hamming distance for 32-byte binary descriptors
    (uint8), L2 for float32 descriptors of up to 128 dimensions (what the reference's BFMatcher() does for "SIFT" /
    "SURF", :394-396).  k_best 1 (the trackers' setting, pose_est_tools.py:686) and 2, the "SIFT" k_best == 2 ratio
    rule, radius match for binary descriptors (use_radius_match, at most 512 matches per query).
    matcher_type "FLANN" (:384-393: randomised KD-trees / LSH tables, approximate) is served by the same EXACT searches:
    the neighbours FLANN approximates.  Not built (raises): k_best > 2, radius match on float descriptors."""

    def __init__(self, method, matcher_type, k_best, *args, **kwargs):
        self.feature_detection_method = method
        self.matcher_type = matcher_type
        self.k_best = k_best
        self.FLANN_INDEX_KDTREE = 1
        self.FLANN_INDEX_LSH = 6
        self.MIN_MATCH_COUNT = 10
        self.percentage_good_matches = kwargs.get("percentage_good_matches", 1.0)
        self.num_of_features = kwargs.get("num_of_features", 100)
        self.use_radius_match = kwargs.get("use_radius_match", False)
        if str(matcher_type).upper() not in ("BF", "FLANN"):
            raise ValueError("matcher_type %r: \"BF\" or \"FLANN\"" % matcher_type)
        self._ctx = kwargs.get("context", None)

    def _context(self):
        if self._ctx is None:
            from ..runtime import default_context
            self._ctx = default_context()
        return self._ctx

    @staticmethod
    def _device_desc(ctx, d, name):
        import torch
        if isinstance(d, torch.Tensor):
            t = d
        else:
            a = np.asarray(d)
            if a.dtype != np.uint8:
                raise NotImplementedError("%s: dtype %s -- only uint8 (binary, Hamming) descriptors are built" % (name, a.dtype))
            t = torch.from_numpy(np.ascontiguousarray(a))
        if t.dim() != 2 or t.shape[1] != 32:
            raise ValueError("%s: expected [n, 32] uint8 descriptors, got %s" % (name, tuple(t.shape)))
        n = t.shape[0]
        buf = torch.zeros((1, max(n, 1), 32), dtype=torch.uint8, device=ctx.device)
        if n:
            buf[0, :n] = t.to(ctx.device)
        return buf, n

    def match_arrays_many(self, pairs):
        """[(query_descriptors, train_descriptors), ...] -> [(query_idx, train_idx, distance), ...]: match_arrays for
        every pair, but as ONE batched launch + one copy back when the matcher is in its best-match mode (the per-bucket
        loop of match_features_panoramic_top_bottom costs a dozen launch / synchronise round trips per frame otherwise)."""
        import torch
        if self.use_radius_match or self.k_best != 1 or len(pairs) == 0 or any(self._is_float(q) for q, _ in pairs):
            return [self.match_arrays(q, t) for q, t in pairs]
        ctx = self._context()
        qs = [np.ascontiguousarray(np.asarray(q)) for q, _ in pairs]
        ts = [np.ascontiguousarray(np.asarray(t)) for _, t in pairs]
        for name, arrs in (("query_descriptors", qs), ("train_descriptors", ts)):
            for a in arrs:
                if a.dtype != np.uint8:
                    raise NotImplementedError("%s: dtype %s -- only uint8 (binary, Hamming) descriptors are built" % (name, a.dtype))
                if a.ndim != 2 or a.shape[1] != 32:
                    raise ValueError("%s: expected [n, 32] uint8 descriptors, got %s" % (name, a.shape))
        P = len(pairs)
        nq, nt = np.array([a.shape[0] for a in qs], np.int32), np.array([a.shape[0] for a in ts], np.int32)
        cq, ct = max(1, int(nq.max())), max(1, int(nt.max()))
        hq, ht = np.zeros((P, cq, 32), np.uint8), np.zeros((P, ct, 32), np.uint8)
        for i in range(P):
            hq[i, :nq[i]] = qs[i]
            ht[i, :nt[i]] = ts[i]
        dev = ctx.device
        nq_d, nt_d = torch.from_numpy(nq).to(dev), torch.from_numpy(nt).to(dev)
        keys = ctx.match_hamming(torch.from_numpy(hq).to(dev), torch.from_numpy(ht).to(dev), nq_d, nt_d, k=1)
        order = ctx.sort_matches(keys, nq_d)
        from .._lib import KEY_SHIFT, KEY_IDX_MASK
        kh, oh = keys[:, :, 0].cpu().numpy().astype(np.int64), order.cpu().numpy().astype(np.int64)
        out = []
        for i in range(P):
            if nq[i] == 0 or nt[i] == 0:
                out.append((np.empty(0, np.int64),) * 2 + (np.empty(0, np.float32),))
                continue
            o = oh[i, :nq[i]]
            k = kh[i][o]
            out.append((o, k & KEY_IDX_MASK, (k >> KEY_SHIFT).astype(np.float32)))
        return out

    @staticmethod
    def _is_float(d):
        import torch
        return (d.dtype in (torch.float32, torch.float64)) if isinstance(d, torch.Tensor) else np.asarray(d).dtype.kind == "f"

    def _l2_arrays(self, query_descriptors, train_descriptors):
        """Float descriptors: (query_idx, train_idx, distance) as match_arrays gives them for binary ones."""
        import torch
        if self.use_radius_match:
            raise NotImplementedError("radius match on float descriptors is not built")
        if self.k_best > 2:
            raise NotImplementedError("k_best > 2 is not built")
        ctx = self._context()
        q = np.ascontiguousarray(np.asarray(query_descriptors.cpu() if isinstance(query_descriptors, torch.Tensor) else
                                            query_descriptors), dtype=np.float32)
        t = np.ascontiguousarray(np.asarray(train_descriptors.cpu() if isinstance(train_descriptors, torch.Tensor) else
                                            train_descriptors), dtype=np.float32)
        if q.ndim != 2 or t.ndim != 2 or q.shape[1] != t.shape[1] or not 1 <= q.shape[1] <= 128:
            raise ValueError("float descriptors must be [n, dim] with the same dim <= 128; got %s and %s" % (q.shape, t.shape))
        nq, nt = q.shape[0], t.shape[0]
        if nq == 0 or nt == 0:
            return (np.empty(0, np.int64),) * 2 + (np.empty(0, np.float32),)
        dev = ctx.device
        k = 2 if self.k_best == 2 else 1
        keys = ctx.match_l2(torch.from_numpy(q[None]).to(dev), torch.from_numpy(t[None]).to(dev),
                            torch.tensor([nq], dtype=torch.int32, device=dev), torch.tensor([nt], dtype=torch.int32, device=dev), k=k)
        ctx.synchronize()
        kh = keys[0].cpu().numpy()                                     # [nq, k] int64 (u64 bit patterns)
        dist = (kh >> 32).astype(np.uint32).view(np.float32)
        idx = kh & 0xFFFFFFFF
        none = kh == -1
        ratio_rule = k == 2 and str(self.feature_detection_method).upper() == "SIFT"
        if k == 1 or ratio_rule:
            order = np.argsort(dist[:, 0], kind="stable")            # sorted(matches, key=distance), ties in query order
            qi, ti, di = order, idx[order, 0], dist[order, 0]
            if ratio_rule:
                keep = ~none[order, 1] & (di < dist[order, 1] * np.float32(0.75))
                qi, ti, di = qi[keep], ti[keep], di[keep]
            return qi.astype(np.int64), ti.astype(np.int64), di
        flat_d, flat_i, flat_none = dist.reshape(-1), idx.reshape(-1), none.reshape(-1)   # [q0 best, q0 second, q1 best, ...]
        order = np.argsort(flat_d, kind="stable")
        order = order[~flat_none[order]]
        return (order // 2).astype(np.int64), flat_i[order].astype(np.int64), flat_d[order]

    def match_arrays(self, query_descriptors, train_descriptors, max_descriptor_distance_radius=-1):
        """-> (query_idx, train_idx, distance) numpy arrays in the order of match()."""
        import torch
        if self._is_float(query_descriptors) or self._is_float(train_descriptors):
            return self._l2_arrays(query_descriptors, train_descriptors)
        if self.use_radius_match:
            return self._radius_arrays(query_descriptors, train_descriptors, max_descriptor_distance_radius)
        if self.k_best > 2:
            raise NotImplementedError("k_best > 2 is not built")
        ctx = self._context()
        dq, nq = self._device_desc(ctx, query_descriptors, "query_descriptors")
        dt, nt = self._device_desc(ctx, train_descriptors, "train_descriptors")
        if nq == 0 or nt == 0:
            return (np.empty(0, np.int64),) * 2 + (np.empty(0, np.float32),)
        from .._lib import KEY_SHIFT, KEY_IDX_MASK, KEY_NONE
        dev = ctx.device
        nq_t = torch.tensor([nq], dtype=torch.int32, device=dev)
        nt_t = torch.tensor([nt], dtype=torch.int32, device=dev)
        k = 2 if self.k_best == 2 else 1
        keys = ctx.match_hamming(dq, dt, nq_t, nt_t, k=k)  # [1, nq, k]
        ratio_rule = k == 2 and str(self.feature_detection_method).upper() == "SIFT"
        if k == 1 or ratio_rule:
            # sorted(matches, key=distance) of one match per query (:442-444); for the ratio rule the filter
            # below keeps the sorted order of the survivors, which is the sorted order of the filtered list
            best = keys[:, :, :1].contiguous()
            order = ctx.sort_matches(best, nq_t)
            kh = keys[0, :nq].cpu().numpy().astype(np.int64)
            oh = order[0, :nq].cpu().numpy().astype(np.int64)
            q = oh
            t = kh[oh, 0] & KEY_IDX_MASK
            d = (kh[oh, 0] >> KEY_SHIFT).astype(np.float32)
            if ratio_rule:
                second = kh[oh, 1]
                keep = (second != KEY_NONE) & (d < (second >> KEY_SHIFT).astype(np.float32) * 0.75)
                q, t, d = q[keep], t[keep], d[keep]
            return q, t, d
        # k_best == 2: the k-lists are flattened [q0 best, q0 second, q1 best, ...] and sorted by distance (:439)
        flat = keys.reshape(1, 2 * nq, 1)
        n2 = torch.tensor([2 * nq], dtype=torch.int32, device=dev)
        order = ctx.sort_matches(flat, n2)
        kh = flat[0, :, 0].cpu().numpy().astype(np.int64)
        oh = order[0, :2 * nq].cpu().numpy().astype(np.int64)
        oh = oh[kh[oh] != KEY_NONE]  # a train set of one descriptor gives k-lists of length 1
        return oh // 2, kh[oh] & KEY_IDX_MASK, (kh[oh] >> KEY_SHIFT).astype(np.float32)

    RADIUS_CAP = 512

    def _radius_arrays(self, query_descriptors, train_descriptors, radius):
        """camera_models.py:412-415: radiusMatch, the per-query lists flattened in query order, then sorted by
        distance (stable, :444).  Within one query, equal distances come in train order."""
        import torch
        from .._lib import KEY_SHIFT, KEY_IDX_MASK
        ctx = self._context()
        dq, nq = self._device_desc(ctx, query_descriptors, "query_descriptors")
        dt, nt = self._device_desc(ctx, train_descriptors, "train_descriptors")
        if nq == 0 or nt == 0 or radius < 0:
            return (np.empty(0, np.int64),) * 2 + (np.empty(0, np.float32),)
        dev = ctx.device
        nq_t = torch.tensor([nq], dtype=torch.int32, device=dev)
        nt_t = torch.tensor([nt], dtype=torch.int32, device=dev)
        keys, counts = ctx.match_radius(dq, dt, nq_t, nt_t, int(np.floor(radius)), self.RADIUS_CAP)
        ctx.synchronize()
        k = keys[0, :nq].cpu().numpy().astype(np.int64)
        c = np.minimum(counts[0, :nq].cpu().numpy(), self.RADIUS_CAP)
        q = np.repeat(np.arange(nq, dtype=np.int64), c)
        flat = k[np.arange(self.RADIUS_CAP)[None, :] < c[:, None]]
        d = flat >> KEY_SHIFT
        order = np.argsort(d, kind="stable")
        return q[order], (flat & KEY_IDX_MASK)[order], d[order].astype(np.float32)

    def match(self, query_descriptors, train_descriptors, max_descriptor_distance_radius=-1):
        """-> MatchList (a list of DMatch), ascending by distance, ties in query order."""
        return MatchList(*self.match_arrays(query_descriptors, train_descriptors, max_descriptor_distance_radius))


class RGBDCamModel(object):
    """camera_models.py:756-860: the pinhole RGB-D camera (intrinsics, depth convention, units).  get_XYZ /
    get_depth_Z keep the reference's numpy arithmetic (they pin the GPU path's sosvo_rgbd_backproject /
    sosvo_rgbd_assemble); per-frame feature extraction runs on the GPU through RGBDFrame."""

    def __init__(self, **kwargs):
        self.fx = kwargs.get("fx", 525.0)
        self.fy = kwargs.get("fy", 525.0)
        self.center_x = kwargs.get("center_x", 319.5)
        self.center_y = kwargs.get("center_y", 239.5)
        self.focal_length_m = kwargs.get("focal_length_m", 1.0 / 1000.0)
        self.depth_is_Z = kwargs.get("depth_is_Z", True)
        self.units = kwargs.get("units", "m")
        self.scaling_factor = kwargs.get("scaling_factor", 1. / 1000.0)  # depth PNG counts -> units (:772)
        self.do_undistortion = kwargs.get("do_undistortion", False)
        self.K = np.array([[self.fx, 0, self.center_x], [0, self.fy, self.center_y], [0, 0, 1]])
        self.image_size = kwargs.get("image_size", None)
        self.T_model_wrt_C = np.identity(4)
        self.T_C_wrt_model = np.identity(4)
        self.T_Cest_wrt_Rgt = None
        self.feature_matcher_for_motion = None

    def get_depth_Z(self, depth, uv_coords=None, verbose=False):
        """camera_models.py:781-799: radial depth -> Z (identity when the map already holds Z)."""
        if not self.depth_is_Z:
            if uv_coords is None:
                uv_coords = np.transpose(np.indices(depth.shape[::-1]), (0, 2, 1))
            focal_length = self.focal_length_m
            x_i = (focal_length / self.fx) * (uv_coords[0] - self.center_x)
            y_i = (focal_length / self.fy) * (uv_coords[1] - self.center_y)
            z_i = np.ones_like(x_i) * focal_length
            d_to_img_plane = np.linalg.norm(np.dstack([x_i, y_i, z_i]), axis=-1)
            depth = focal_length * depth / d_to_img_plane
        return depth

    def get_XYZ(self, depth, u_coords=None, v_coords=None):
        """camera_models.py:835-860: XYZ at the given integer pixels ([1, n, 3]) or for the whole map; zero depth -> NaN."""
        depth = self.get_depth_Z(depth=depth, uv_coords=None)
        Z = np.where(depth != 0, depth, np.nan)
        if u_coords is None or v_coords is None:
            uv_coords = np.transpose(np.indices(depth.shape[::-1]), (0, 2, 1))
            u_coords, v_coords = uv_coords[0], uv_coords[1]
        else:
            u_coords, v_coords = u_coords.ravel(), v_coords.ravel()
            Z = Z[v_coords, u_coords]
        X = (u_coords - self.center_x) * Z / self.fx
        Y = (v_coords - self.center_y) * Z / self.fy
        return np.dstack((X, Y, Z))

    # ---- device side ---------------------------------------------------------------------------------------
    def _context(self):
        from ..runtime import default_context
        return default_context()

    def _front_end(self, shape, num_of_features, median_win_size, min_range, max_range, mask):
        """One-frame RGBDFrontEnd per (image shape, detector budget, filter) -- buffers are reused across frames."""
        from ..pipeline import RGBDCamConfig, RGBDFrontEnd
        key = (tuple(shape), int(num_of_features), int(median_win_size), float(min_range), float(max_range),
               None if mask is None else id(mask))
        cache = self.__dict__.setdefault("_front_ends", {})
        if key not in cache:
            cfg = RGBDCamConfig(self.fx, self.fy, self.center_x, self.center_y, self.focal_length_m, self.depth_is_Z,
                                min_range, max_range)
            cache[key] = RGBDFrontEnd(self._context(), cfg, 1, image_shape=shape, num_of_features=num_of_features,
                                      median_win_size=median_win_size, mask=mask)
        return cache[key]
