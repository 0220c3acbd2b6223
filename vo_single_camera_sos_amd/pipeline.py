"""Batched frame-pair front end: the order of operations of one reference VO step
(StereoPanoramicFrame.__init__ -> TrackerStereoSE3.track_frame, omnistereo/pose_est_tools.py:271-402,
:736-847) for B independent frame pairs at once, every buffer resident in HBM, every stage one or two
asynchronous C-ABI calls on one stream, no host synchronisation inside a step.

Pair i tracks frame 2i+1 (current) against frame 2i (reference / keyframe)."""
import numpy as np
import torch

from .device import Context, make_rig


class RigConfig(object):
    """What the hot path needs to know about one calibrated omnistereo rig (host numbers)."""

    def __init__(self, pano_top, pano_bot, F_top, F_bot, min_range, max_range, stereo_min_disp=1.0,
                 stereo_max_hdiff=2.5, f2f_max_hdiff=None, pct_good_matches=1.0):
        self.pano_top, self.pano_bot = tuple(pano_top), tuple(pano_bot)
        self.F_top, self.F_bot = np.asarray(F_top, np.float64), np.asarray(F_bot, np.float64)
        self.min_range, self.max_range = float(min_range), float(max_range)
        self.stereo_min_disp, self.stereo_max_hdiff = float(stereo_min_disp), float(stereo_max_hdiff)
        # TrackerStereoSE3.bootstrap_tracker (pose_est_tools.py:866): 0.125 * 0.5 * pano cols
        self.f2f_max_hdiff = 0.125 * 0.5 * self.pano_top[0] if f2f_max_hdiff is None else float(f2f_max_hdiff)
        self.pct_good_matches = float(pct_good_matches)

    def as_struct(self):
        return make_rig(self.pano_top, self.pano_bot, self.F_top, self.F_bot, self.min_range, self.max_range,
                        self.stereo_min_disp, self.stereo_max_hdiff, self.f2f_max_hdiff, self.pct_good_matches)


class FramePairPipeline(object):
    def __init__(self, ctx, rig, n_pairs, nmask=12, bucket_cap=192, frame_cap=2048, thr=None, max_iter=2000,
                 adaptive=False, seed=0, lm_iter=30, front_end=None, ransac_solver="P3P"):
        """front_end: an ImageFrontEnd over 2 * n_pairs frames; its keypoint/descriptor buffers are used in
        place (no copy).  Without it, keypoints are loaded with load_keypoints().
        ransac_solver: "GP3P" = generalised P3P on samples across both mirrors (what the reference's non-central RANSAC
        uses, pose_est_tools.py:696); "P3P" = the three solve points of a sample from one mirror (BASELINE config 2)."""
        assert isinstance(ctx, Context)
        self.ctx, self.rig_cfg, self.rig = ctx, rig, rig.as_struct()
        self.front_end = front_end
        if front_end is not None:
            assert front_end.F == 2 * int(n_pairs)
            nmask, bucket_cap = front_end.model.nmask, front_end.kp_cap
        self.B, self.F, self.NM = int(n_pairs), 2 * int(n_pairs), int(nmask)
        self.bucket_cap, self.frame_cap, self.corr_cap = int(bucket_cap), int(frame_cap), 2 * int(frame_cap)
        # TrackerSE3.set_global_parameters_for_tracking (pose_est_tools.py:675-676)
        self.thr = float(1.0 - np.cos(np.deg2rad(5.0))) if thr is None else float(thr)
        self.max_iter, self.adaptive, self.seed, self.lm_iter = int(max_iter), bool(adaptive), int(seed), int(lm_iter)
        if str(ransac_solver).upper() not in ("P3P", "GP3P"):
            raise ValueError("ransac_solver: P3P or GP3P")
        self.gp3p = str(ransac_solver).upper() == "GP3P"
        dev = ctx.device
        P = self.F * self.NM
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)  # noqa: E731
        if front_end is not None:
            v = front_end.view_arrays()
            self.kp_top, self.kp_bot, self.desc_top, self.desc_bot = v["kp_top"], v["kp_bot"], v["desc_top"], v["desc_bot"]
            self.n_top, self.n_bot = v["n_top"], v["n_bot"]
        else:
            self.kp_top, self.kp_bot = z((P, bucket_cap, 2), torch.float32), z((P, bucket_cap, 2), torch.float32)
            self.desc_top, self.desc_bot = z((P, bucket_cap, 32), torch.uint8), z((P, bucket_cap, 32), torch.uint8)
            self.n_top, self.n_bot = z((P,), torch.int32), z((P,), torch.int32)
        self.s_keys, self.s_order = z((P, bucket_cap, 1), torch.uint32), z((P, bucket_cap), torch.int32)
        Fc, Cc = self.frame_cap, self.corr_cap
        # descriptors and counts of both views back to back (view-major): the two frame-to-frame matchings of a step
        # (top with top, bottom with bottom) are ONE launch of 2B problems, and so are the two sorts
        self._d2 = z((2, self.F, Fc, 32), torch.uint8)
        self._M2 = z((2, self.F), torch.int32)
        self.frames = dict(m_top=z((self.F, Fc, 2), torch.float32), m_bot=z((self.F, Fc, 2), torch.float32),
                           d_top=self._d2[0], d_bot=self._d2[1],
                           X=z((self.F, Fc, 3), torch.float64), b_top=z((self.F, Fc, 3), torch.float64),
                           b_bot=z((self.F, Fc, 3), torch.float64), M=self._M2[0],
                           n_cand=z((self.F,), torch.int32))
        self.ref_frame = torch.arange(0, self.F, 2, dtype=torch.int32, device=dev)
        self.cur_frame = torch.arange(1, self.F, 2, dtype=torch.int32, device=dev)
        self._q2 = torch.cat([self.cur_frame, self.cur_frame + self.F])   # problem i: top view, B + i: bottom view
        self._t2 = torch.cat([self.ref_frame, self.ref_frame + self.F])
        self._k2, self._o2 = z((2 * self.B, Fc, 1), torch.uint32), z((2 * self.B, Fc), torch.int32)
        self.k_top, self.k_bot = self._k2[: self.B], self._k2[self.B:]
        self.o_top, self.o_bot = self._o2[: self.B], self._o2[self.B:]
        self.corr = dict(f=z((self.B, Cc, 3), torch.float64), p=z((self.B, Cc, 3), torch.float64),
                         cam=z((self.B, Cc), torch.int32), q=z((self.B, Cc), torch.int32),
                         t=z((self.B, Cc), torch.int32), n=z((self.B,), torch.int32), n_top=z((self.B,), torch.int32))
        self.ransac = dict(T=z((self.B, 3, 4), torch.float64), mask=z((self.B, Cc), torch.uint8),
                           idx=z((self.B, Cc), torch.int32), n_inliers=z((self.B,), torch.int32),
                           info=z((self.B, 4), torch.int32))
        self.T = z((self.B, 3, 4), torch.float64)
        self.lm_cost, self.lm_iters = z((self.B,), torch.float64), z((self.B,), torch.int32)
        self.cam_off = torch.from_numpy(np.stack([rig.F_top, rig.F_bot])).to(dev)
        self.cam_rot = torch.from_numpy(np.stack([np.eye(3), np.eye(3)])).to(dev)

    # ---- inputs ------------------------------------------------------------------------------
    def load_keypoints(self, packed):
        """Image-free entry: per-bucket keypoints/descriptors as numpy arrays (tests/synth.pack_buckets)."""
        dev = self.ctx.device
        for name in ("kp_top", "kp_bot", "desc_top", "desc_bot", "n_top", "n_bot"):
            getattr(self, name).copy_(torch.from_numpy(packed[name]).to(dev))

    # ---- one step ----------------------------------------------------------------------------
    def stereo(self):
        c = self.ctx
        c.match_hamming(self.desc_bot, self.desc_top, self.n_bot, self.n_top, k=1, keys=self.s_keys)
        c.sort_matches(self.s_keys, self.n_bot, order=self.s_order)
        c.stereo_assemble(self.rig, self.kp_top, self.kp_bot, self.desc_top, self.desc_bot, self.n_top, self.n_bot,
                          self.s_keys, self.s_order, self.F, self.NM, self.frame_cap, out=self.frames)

    def track(self):
        c, fr = self.ctx, self.frames
        self._M2[1].copy_(self._M2[0])                      # the bottom view's blocks have the same counts
        d2, M2 = self._d2.view(2 * self.F, self.frame_cap, 32), self._M2.view(2 * self.F)
        c.match_hamming(d2, d2, M2, M2, k=1, keys=self._k2, q_slot=self._q2, t_slot=self._t2)   # both views, 2B problems
        c.sort_matches(self._k2, M2, order=self._o2, q_slot=self._q2)
        c.f2f_assemble(self.rig, fr, self.ref_frame, self.cur_frame, self.k_top, self.o_top, self.k_bot, self.o_bot,
                       self.corr_cap, out=self.corr)
        co = self.corr
        c.ransac_abs_pose(co["f"], co["p"], co["n"], self.thr, self.max_iter, seed=self.seed, adaptive=self.adaptive,
                          cam=co["cam"], cam_off=self.cam_off, cam_rot=self.cam_rot, cam_rot_identity=True,
                          out=self.ransac, gp3p=self.gp3p)
        self.T.copy_(self.ransac["T"])
        c.refine_abs_pose(co["f"], co["p"], co["n"], self.T, idx=self.ransac["idx"], m=self.ransac["n_inliers"],
                          cam=co["cam"], cam_off=self.cam_off, cam_rot=self.cam_rot, max_lm_iter=self.lm_iter,
                          cost=self.lm_cost, iters=self.lm_iters)

    def step(self):
        """The whole hot path for all B pairs (asynchronous): image front end (when attached), stereo
        correspondences, frame-to-frame tracking."""
        if self.front_end is not None:
            self.front_end.run()
        self.stereo()
        self.track()
        return self.T

    def results(self, out=None):
        """[B,16] f64 rows: 3x4 refined pose, n_inliers, n_correspondences, status, RANSAC best iteration
        (the per-pair record gathered over ranks in the multi-GPU configuration)."""
        if out is None:
            out = torch.empty((self.B, 16), dtype=torch.float64, device=self.ctx.device)
        out[:, :12] = self.T.reshape(self.B, 12)
        out[:, 12] = self.ransac["n_inliers"].to(torch.float64)
        out[:, 13] = self.corr["n"].to(torch.float64)
        out[:, 14] = self.ransac["info"][:, 2].to(torch.float64)
        out[:, 15] = self.ransac["info"][:, 0].to(torch.float64)
        return out


class OverlappedFramePairs(object):
    """The hot path for B pairs split over S HIP streams (one libsosvo context, front end and pipeline per
    stream, model constants shared).  The path alternates between a VALU-bound stage (K1-K3, the median) and
    latency-bound ones (corner selection, descriptors, matching of small problems, RANSAC bookkeeping, LM) that
    leave most of the chip's issue slots idle; a token (HIP event) passed from stream to stream serialises the
    median launches, so that while one part of the batch is in its median the other parts' latency-bound
    kernels fill the gaps.  Results are those of one FramePairPipeline over all B pairs, bit for bit (pair i
    samples with seed + i)."""

    class _Part(object):
        pass

    def __init__(self, device, gums, omni_shape, rig, n_pairs, n_streams=2, num_of_features=1000, kp_cap=512,
                 frame_cap=2048, max_iter=2000, adaptive=False, seed=0, detection_method="GFT", lm_iter=30, thr=None,
                 serialize_medians=True, ransac_solver="P3P", median_win_size=11):
        """median_win_size: StereoPanoramicFrame.median_win_size (11, pose_est_tools.py:296); 0 = no median blur (the
        RGB-D frames' setting, :427) -- then the colour panoramas are materialised and only converted to gray."""
        from .frontend import DeviceImageModel, ImageFrontEnd
        from .parallel import shard_range
        self.main = Context(device)                       # model constants + result collection: torch's current stream
        self.device = self.main.device
        self.model = DeviceImageModel(self.main, gums, omni_shape)
        m = self.model
        m.unwrap_table = self.main.unwrap_prepare(m.omni_masks, m.map_x, m.map_y, (m.H, m.W))
        self.B, self.S = int(n_pairs), max(1, min(int(n_streams), int(n_pairs)))
        self.out = torch.zeros((self.B, 16), dtype=torch.float64, device=self.device)
        self.main.synchronize()
        self.parts = []
        for s in range(self.S):
            lo, hi = shard_range(self.B, s, self.S)
            p = self._Part()
            p.lo, p.hi = lo, hi
            p.stream = torch.cuda.Stream(self.device)
            with torch.cuda.stream(p.stream):
                p.ctx = Context(self.device.index, p.stream)
                if self.S > 1:
                    p.ctx.set_hint_shared_device(True)   # the parts run side by side: leave wave slots for each other
                p.fe = ImageFrontEnd(p.ctx, m, 2 * (hi - lo), detection_method=detection_method,
                                     num_of_features=num_of_features, kp_cap=kp_cap, keep_panoramas=False,
                                     median_win_size=median_win_size)
                p.pipe = FramePairPipeline(p.ctx, rig, hi - lo, frame_cap=frame_cap, max_iter=max_iter, adaptive=adaptive,
                                           seed=seed + lo, front_end=p.fe, lm_iter=lm_iter, thr=thr,
                                           ransac_solver=ransac_solver)
            p.median_done = torch.cuda.Event()
            p.done = torch.cuda.Event()
            self.parts.append(p)
        self.thr = self.parts[0].pipe.thr
        self.use_token = bool(serialize_medians)
        self._token = None       # the last median launch
        self._consumed = None    # self.out has been read by its consumer (see results())
        torch.cuda.synchronize(self.device)

    def contexts(self):
        return [p.ctx for p in self.parts]

    def load_frames(self, omni):
        """omni [2B,H,W,3] u8 (numpy or torch): pair i = frames 2i (reference) and 2i+1 (current)."""
        for p in self.parts:
            with torch.cuda.stream(p.stream):
                p.fe.load_frames(omni[2 * p.lo:2 * p.hi])
        torch.cuda.synchronize(self.device)

    def step_from_host(self, omni_pinned):
        """One pass over all B pairs whose omni frames are handed over in (pinned) HOST memory [2B,H,W,3] u8: every part
        copies its frames on the engine's copy stream into one of two device buffers (the copy for step k+1 overlaps the
        kernels of step k; a buffer is reused once the median that read it has finished) and then runs step()'s
        sequence.  This is the PCIe-inclusive form of the hot path."""
        k = self._host_step = getattr(self, "_host_step", -1) + 1
        j = k & 1
        if not hasattr(self, "_copy_streams"):
            # TWO copy streams shared by all parts (part i copies on stream i % 2): measured host-to-device rates with
            # three parts -- one queue 46 GB/s, two 55 GB/s, three 35 GB/s
            self._copy_streams = [torch.cuda.Stream(self.device) for _ in range(min(2, self.S))]
        for i, p in enumerate(self.parts):
            cs = self._copy_streams[i % len(self._copy_streams)]
            if not hasattr(p, "omni_bufs"):
                p.omni_bufs = [p.fe.omni, torch.empty_like(p.fe.omni)]
                p.copied = [torch.cuda.Event(), torch.cuda.Event()]
                # buffer 0 is the resident one: a preceding step()'s median on p.stream may still be reading it
                busy = torch.cuda.Event()
                busy.record(p.stream)
                p.omni_free = [busy, None]
            with torch.cuda.stream(cs):
                if p.omni_free[j] is not None:
                    cs.wait_event(p.omni_free[j])
                p.omni_bufs[j].copy_(omni_pinned[2 * p.lo:2 * p.hi], non_blocking=True)
                p.copied[j].record(cs)
            p.fe.omni = p.omni_bufs[j]
            p.stream.wait_event(p.copied[j])
        self.step(_after_median=lambda p: self._mark_free(p, j))

    @staticmethod
    def _mark_free(p, j):
        ev = torch.cuda.Event()
        ev.record(p.stream)
        p.omni_free[j] = ev

    def step(self, _after_median=None):
        """One pass of the whole hot path over all B pairs (asynchronous)."""
        for p in self.parts:
            with torch.cuda.stream(p.stream):
                if self._token is not None and self.use_token:
                    p.stream.wait_event(self._token)      # medians take turns
                p.fe.run_images()
                p.median_done.record(p.stream)
                self._token = p.median_done
                if _after_median is not None:
                    _after_median(p)
                p.fe.run_features()
                p.pipe.stereo()
                p.pipe.track()
                if self._consumed is not None:
                    p.stream.wait_event(self._consumed)   # the previous step's records have been read
                p.pipe.results(out=self.out[p.lo:p.hi])
                p.done.record(p.stream)

    def capture_graph(self):
        """One step() + results() of ALL parts captured into a torch.cuda.CUDAGraph; replay() then re-runs the whole
        hot path on the frames resident in the parts' buffers and leaves the records in self.out.
        What a correct capture needs: the parts' streams must JOIN the capture -- each waits on
        an event recorded on the capturing stream -- otherwise their kernels run eagerly, outside the graph, and a replay
        returns whatever the buffers hold; no wait may refer to an event recorded before the capture (the median token
        and the "records consumed" event of an earlier eager step); the per-kernel profile must be off (its event pairs
        would be captured and re-recorded); the library's scratch must have its final size (one eager step first)."""
        torch.cuda.synchronize(self.device)
        self.profile_enable(False)
        self._token, self._consumed = None, None
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            cur = torch.cuda.current_stream(self.device)
            fork = torch.cuda.Event()
            fork.record(cur)
            for p in self.parts:
                p.stream.wait_event(fork)
            self.step()
            self.results()          # the capturing stream waits for every part: the join
        self._token, self._consumed = None, None   # events recorded while capturing are nodes of the graph, not events
        torch.cuda.synchronize(self.device)
        return g

    def results(self):
        """[B,16] f64 records of the last step, valid on torch's current stream (which is made to wait for every
        part).  Call consumed() after enqueuing whatever reads them, before the next step()."""
        cur = torch.cuda.current_stream(self.device)
        for p in self.parts:
            cur.wait_event(p.done)
        return self.out

    def consumed(self):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._consumed = ev

    def profile_enable(self, on=True):
        for p in self.parts:
            p.ctx.profile_enable(on)

    def profile_read(self):
        out = []
        for p in self.parts:
            out.extend(p.ctx.profile_read())
        return out

    def close(self):
        torch.cuda.synchronize(self.device)
        for p in self.parts:
            p.ctx.close()
        self.main.close()


class FramePairBatch(object):
    """The same hot path behind ONE C-ABI call per step (sosvo_frame_pair_batch): what a non-Python host would
    bind.  `model` is a DeviceImageModel (model constants in HBM); the workspace is allocated once."""

    def __init__(self, ctx, model, rig, n_pairs, num_of_features=1000, kp_cap=None, frame_cap=2048, median_win_size=11,
                 quality=0.01, min_distance=5.0, edge=31, thr=None, max_iter=2000, adaptive=False, seed=0, lm_iter=30,
                 n_streams=1, ransac_solver="P3P"):
        """n_streams > 1: sosvo_frame_pair_batch_streams (the batch split over internal HIP streams of the library)."""
        from . import _lib, orb_pattern
        self.ctx, self.model, self.rig_cfg, self.rig = ctx, model, rig, rig.as_struct()
        self.B, self.n_streams = int(n_pairs), max(1, min(int(n_streams), int(n_pairs)))
        if kp_cap is None:
            kp_cap = int(min(1024, max(64, -(-int(num_of_features) // 64) * 64)))
        cos_a, sin_a = orb_pattern.angle_cos_sin(orb_pattern.GFT_KEYPOINT_ANGLE)
        c = _lib.BatchCfg()
        c.n_pairs, c.H, c.W, c.rows, c.cols, c.nmask = self.B, model.H, model.W, model.rows, model.cols, model.nmask
        c.kp_cap, c.frame_cap, c.median_ksize, c.max_corners, c.edge = int(kp_cap), int(frame_cap), int(median_win_size), \
            int(num_of_features), int(edge)
        c.ransac_max_iter, c.ransac_adaptive, c.lm_max_iter = int(max_iter), 1 if adaptive else 0, int(lm_iter)
        c.quality, c.min_distance = float(quality), float(min_distance)
        c.ransac_threshold = float(1.0 - np.cos(np.deg2rad(5.0))) if thr is None else float(thr)
        c.seed, c.cos_a, c.sin_a = int(seed), float(cos_a), float(sin_a)
        c.ransac_flags, c.reserved = (_lib.FLAG_GP3P if str(ransac_solver).upper() == "GP3P" else 0), 0
        self.cfg = c
        if getattr(model, "unwrap_table", None) is None:  # once per model
            model.unwrap_table = ctx.unwrap_prepare(model.omni_masks, model.map_x, model.map_y, (model.H, model.W))
        nbytes = ctx.frame_pair_batch_workspace(c) if self.n_streams == 1 else ctx.frame_pair_batch_streams_workspace(c, self.n_streams)
        if nbytes <= 0:
            raise ValueError("bad batch configuration")
        self.workspace = torch.empty((nbytes,), dtype=torch.uint8, device=ctx.device)
        self.omni = torch.zeros((2 * self.B, model.H, model.W, 3), dtype=torch.uint8, device=ctx.device)
        self.out = torch.zeros((self.B, 16), dtype=torch.float64, device=ctx.device)

    def load_frames(self, omni):
        t = torch.from_numpy(np.ascontiguousarray(omni)) if isinstance(omni, np.ndarray) else omni
        self.omni.copy_(t.to(self.ctx.device))

    def step(self):
        """-> results [B,16] f64 (asynchronous)."""
        m = self.model
        return self.ctx.frame_pair_batch(self.rig, self.cfg, self.omni, m.unwrap_table, m.mask_bits, m.pattern, self.workspace,
                                         results=self.out, n_streams=self.n_streams)

    def enqueue(self, results=None):
        """A step WITHOUT the join (sosvo_frame_pair_batch_streams_enqueue, n_streams > 1): consecutive calls overlap on
        the library's internal streams.  Pass alternating `results` buffers while something still reads the previous
        step's records; call join() before reading records on the context's stream."""
        m = self.model
        return self.ctx.frame_pair_batch(self.rig, self.cfg, self.omni, m.unwrap_table, m.mask_bits, m.pattern, self.workspace,
                                         results=self.out if results is None else results, n_streams=self.n_streams,
                                         join=self.n_streams <= 1)

    def join(self):
        if self.n_streams > 1:
            self.ctx.frame_pair_batch_join()

    def results(self):
        return self.out


_STAGE_POOL = None
_STAGE_THREADS = 4


def _stage_rows(dst_np, images):
    """Frames of a window into a pinned staging buffer, four numpy copies at a time (numpy releases the GIL inside a large
    copy: 11.6 GB/s against 6.2 for one thread on the build container's cores -- a window of RGB-D frames is 69 MB.  torch's
    multi-threaded copy_ into PINNED memory, 26 GB/s here on pageable memory, ran at 0.5 GB/s on the GPU box)."""
    global _STAGE_POOL
    if len(images) < 4:
        for i, im in enumerate(images):
            np.copyto(dst_np[i], im)
        return
    if _STAGE_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _STAGE_POOL = ThreadPoolExecutor(_STAGE_THREADS, thread_name_prefix="sosvo-stage")
    list(_STAGE_POOL.map(lambda i: np.copyto(dst_np[i], images[i]), range(len(images))))


class _SequenceBase(object):
    """Window / slot bookkeeping shared by the sequence engines (include/sosvo.h "Sequence mode"): two window halves
    [0, W) and [W, 2W) used alternately (the last frame of the previous window stays readable) and slot 2W for the current
    keyframe.  A tracking record is a pure function of (reference frame, current frame, seed): push_window() tracks every
    frame of the window against its PREDECESSOR in one batched call; the VO loop takes that record where the predecessor
    was the reference and asks for one serial track() call where it was not.  Either way the records are those of the
    serial loop, whatever the window size.  Subclasses provide _stage / _front_end / _track / _counts / _copy."""

    early_upload = True   # run_VO's staging thread starts a window's host-to-device copy itself (upload_staged); False: A/B
    enqueue_ahead = True  # run_VO enqueues window k + 1 before it works through window k's records (enqueue_staged); False: A/B

    def _init_windows(self, window):
        self.W = max(1, int(window))
        self.slots, self.key_slot = 2 * self.W + 1, 2 * self.W
        self.reset()

    def reset(self):
        """Forget the sequence (an engine is reused from run to run: its buffers -- pinned host memory among them -- are the
        expensive part of its construction)."""
        self.half = 1            # the half the NEXT window goes to is 1 - half
        self.last_slot = None    # slot of the newest frame of the sequence
        self.last2_slot = None   # ... and of the one before it
        self.frames_seen = 0     # frames pushed so far (frame t of the sequence is tracked with seed t - 1)
        self.serial_calls = 0    # tracking calls the speculation did not cover
        self._key_src = None     # slot of a promoted frame whose record has not been copied to the keyframe slot yet
        if getattr(self, "_pending", None) is not None:
            self.ctx.synchronize()   # (a window an aborted run left on the stream: let it finish before its buffers are reused)
        self._pending = None     # the window enqueue_staged() put on the stream and collect() has not fetched yet
        self._up_pending = [0, 0]  # frames of pinned buffer b already on their way to device buffer b (upload_staged)
        if not hasattr(self, "_copy_stream"):
            self._copy_stream, self._up_event, self._win_events = None, None, None
        elif self._copy_stream is not None:
            self._copy_stream.synchronize()   # (a copy an aborted run left in flight)
        # host wall clock per stage; gpu_windows: the windows' own time on the stream (events around enqueue_staged's work)
        self.stage_s = dict(stage_to_pinned=0.0, enqueue=0.0, wait_and_readback=0.0, serial_track=0.0, gpu_windows=0.0)

    def push_window(self, images):
        """images: list of n <= window frames that continue the sequence (omnistereo: omni images [H,W,3] u8; RGB-D:
        (bgr, depth) tuples).  -> list of n dicts(slot, count = num_valid_keypoints, seed, spec_ref = slot the speculative
        record was tracked against or None, spec = that [16] record (numpy) or None).  One host synchronisation."""
        import time
        n = len(images)
        if n == 0:
            return []
        if n > self.W:
            raise ValueError("more frames than the window holds")
        t0 = time.perf_counter()
        self.stage_host(images, 0)
        self.stage_s["stage_to_pinned"] += time.perf_counter() - t0
        return self.push_staged(0, n)

    def stage_host(self, images, buf):
        """Copies the frames into pinned host buffer `buf` (0 / 1).  Touches nothing else: a helper thread may stage the NEXT
        window into the other buffer while this one is processed (numpy's copy releases the GIL) -- run_VO does."""
        self._stage_host(images, int(buf))

    def upload_staged(self, buf, n):
        """Starts the host-to-device copy of the n frames staged in pinned buffer `buf` on the engine's COPY stream, into
        device input buffer `buf`; push_staged(buf, n) then only waits for its event.  Optional: without it push_staged
        copies on the compute stream itself.  The helper thread that staged the window calls it (run_VO does), so the copy
        of window k + 1 runs on the DMA engines while window k's kernels run -- ~0.6 ms of 2.6 ms per window of 32 frames
        off the GPU-side critical path.  Buffer `buf` (host and device side) is free again when the push_staged that
        consumed it has returned (it synchronises)."""
        import torch
        buf, n = int(buf), int(n)
        if n <= 0:
            return
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=self.ctx.device)
            self._up_event = [torch.cuda.Event(), torch.cuda.Event()]
        with torch.cuda.stream(self._copy_stream):
            self._upload(buf, n)
            self._up_event[buf].record(self._copy_stream)
        self._up_pending[buf] = n

    def push_staged(self, buf, n):
        """push_window for n frames already staged in pinned buffer `buf`: enqueue_staged + collect."""
        if n == 0:
            return []
        self.enqueue_staged(buf, n)
        return self.collect()

    def enqueue_staged(self, buf, n):
        """The device work of push_staged(buf, n) WITHOUT waiting for it: copy (or the wait for upload_staged's copy), front end,
        the two speculative tracking batches.  collect() synchronises and returns the window's records.  A caller that has
        the next window ready enqueues it BEFORE it works through the records of the current one (run_VO does): the GPU
        then runs window k + 1 while the host's keyframe policy runs on window k.  What keeps that safe: the window halves
        alternate (window k + 1 refills the half of window k - 1), a promoted frame's record is copied to the keyframe slot
        before its half is refilled (here, as ever), serial track() calls read the keyframe slot and a slot of window k
        only, and everything is ordered on one stream.  One window may be pending."""
        import time
        if n <= 0 or n > self.W:
            raise ValueError("1 .. window frames per push")
        if self._pending is not None:
            raise RuntimeError("collect() the pending window first")
        t1 = time.perf_counter()
        buf = int(buf)
        if self._win_events is None:
            import torch
            self._win_events = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        self.half = 1 - self.half
        first = self.half * self.W
        if self._key_src is not None and first <= self._key_src < first + self.W:
            self._flush_promotion()   # the promoted frame's record is about to be overwritten
        self._win_events[0].record(self.ctx.stream)
        if self._up_pending[buf] == n:            # copied ahead by upload_staged: order the compute stream behind it
            self.ctx.stream.wait_event(self._up_event[buf])
        else:
            self._upload(buf, n)
        self._up_pending[buf] = 0
        self._front_end(n, first, buf)
        # speculative tracking: frame i of the window against its predecessor (the first one against the previous
        # window's last frame; the very first frame of the sequence has nothing to track against)
        slots = [first + i for i in range(n)]
        prev = ([self.last_slot] if self.last_slot is not None else []) + slots[:-1]
        cur = slots if self.last_slot is not None else slots[1:]
        seed0 = self.frames_seen - 1 if self.last_slot is not None else 0   # frame t tracks with seed t - 1
        if cur:
            self._track(prev, cur, seed0, self.spec)
        # second guess: frame t against frame t - 2 -- the reference whenever frame t - 1 was NOT promoted to keyframe (then
        # the keyframe usually is t - 2: a frame is passed over when it has moved too little, rarely twice in a row).  Same
        # seeds (t - 1), so these are the records the serial call would return; what neither guess covers stays serial.
        # (a frame of the previous window is still in the store unless its slot lies in the half being refilled: window 1)
        alive = lambda s_: s_ is not None and not (first <= s_ < first + self.W)  # noqa: E731
        older = ([self.last2_slot] if alive(self.last2_slot) and alive(self.last_slot) else []) + \
            ([self.last_slot] if alive(self.last_slot) else []) + slots                       # frames ... t0-2, t0-1, t0, ...
        k0 = len(older) - n                                     # position of the window's first frame in `older`
        cur2 = [slots[i] for i in range(n) if k0 + i - 2 >= 0]
        prev2 = [older[k0 + i - 2] for i in range(n) if k0 + i - 2 >= 0]
        if cur2:
            self._track(prev2, cur2, self.frames_seen + (n - len(cur2)) - 1, self.spec2)
        self._win_events[1].record(self.ctx.stream)
        self._pending = dict(first=first, n=n, slots=slots, prev=prev, n_cur=len(cur), n_cur2=len(cur2),
                             frames_seen=self.frames_seen, had_last=self.last_slot is not None)
        self.frames_seen += n
        self.last2_slot = slots[-2] if n >= 2 else self.last_slot
        self.last_slot = slots[-1]
        self.stage_s["enqueue"] += time.perf_counter() - t1

    def collect(self):
        """Waits for the window enqueue_staged() put on the stream and returns its n dicts (see push_window)."""
        import time
        p = self._pending
        if p is None:
            raise RuntimeError("no window pending")
        t2 = time.perf_counter()
        n, slots, prev = p["n"], p["slots"], p["prev"]
        counts = self._counts(p["first"], n)   # synchronises
        spec = self.spec[:p["n_cur"]].cpu().numpy() if p["n_cur"] else np.zeros((0, 16))
        spec2 = self.spec2[:p["n_cur2"]].cpu().numpy() if p["n_cur2"] else np.zeros((0, 16))
        self._pending = None
        self.stage_s["wait_and_readback"] += time.perf_counter() - t2
        self.stage_s["gpu_windows"] += 1e-3 * self._win_events[0].elapsed_time(self._win_events[1])
        out = []
        for i in range(n):
            t = p["frames_seen"] + i            # index of the frame in the sequence
            j = i if p["had_last"] else i - 1
            has = j >= 0 and p["n_cur"] > 0
            j2 = i - (n - p["n_cur2"])
            out.append(dict(slot=slots[i], count=counts[i], seed=max(t - 1, 0), spec_ref=prev[j] if has else None,
                            spec=spec[j].copy() if has else None, spec2=spec2[j2].copy() if j2 >= 0 else None))
        return out

    def track(self, ref_slot, cur_slot, seed):
        """One serial tracking call (reference slot, current slot, seed) -> [16] record (numpy); synchronises."""
        import time
        t0 = time.perf_counter()
        self.serial_calls += 1
        if int(ref_slot) == self.key_slot:
            self._flush_promotion()
        self._track([int(ref_slot)], [int(cur_slot)], int(seed), self.one)
        rec = self.one.cpu().numpy()[0]
        self.stage_s["serial_track"] += time.perf_counter() - t0
        return rec

    def promote(self, slot):
        """The frame in `slot` becomes the keyframe.  Its record moves to the keyframe slot LAZILY: when a serial tracking call
        reads that slot, or when the window half the frame lives in is about to be refilled -- most keyframes are replaced
        by the next one before either happens (242 of 256 frames of the bench sequence are promoted, 14 copies are made)."""
        self._key_src = int(slot)

    def _flush_promotion(self):
        if self._key_src is not None:
            self._copy(self._key_src, self.key_slot)
            self._key_src = None


class SequenceEngine(_SequenceBase):
    """Sequence mode of the SOS hot path (the reference's VO loop, run_VO, pose_est_tools.py:1416-1628): the front end of
    every frame is computed ONCE, `window` frames per batch, into a frame store in HBM, and frames are tracked against
    keyframes by slot number (see _SequenceBase)."""

    def __init__(self, ctx, model, rig, window=32, num_of_features=1000, kp_cap=None, frame_cap=2048, median_win_size=11,
                 quality=0.01, min_distance=5.0, edge=31, thr=None, max_iter=210, adaptive=True, lm_iter=30,
                 ransac_solver="GP3P"):
        """Defaults: the trackers' settings (pose_est_tools.py:672-707: 5-degree threshold, the iteration budget of
        compute_num_of_iterations_RANSAC = 210 with the adaptive stop, generalised P3P; LM 30 iterations)."""
        from . import _lib, orb_pattern
        assert isinstance(ctx, Context)
        self.ctx, self.model, self.rig_cfg, self.rig = ctx, model, rig, rig.as_struct()
        self._init_windows(window)
        if kp_cap is None:
            kp_cap = int(min(1024, max(64, -(-int(num_of_features) // 64) * 64)))
        cos_a, sin_a = orb_pattern.angle_cos_sin(orb_pattern.GFT_KEYPOINT_ANGLE)
        c = _lib.BatchCfg()
        c.n_pairs, c.H, c.W, c.rows, c.cols, c.nmask = self.W, model.H, model.W, model.rows, model.cols, model.nmask
        c.kp_cap, c.frame_cap, c.median_ksize, c.max_corners, c.edge = int(kp_cap), int(frame_cap), int(median_win_size), \
            int(num_of_features), int(edge)
        c.ransac_max_iter, c.ransac_adaptive, c.lm_max_iter = int(max_iter), 1 if adaptive else 0, int(lm_iter)
        c.quality, c.min_distance = float(quality), float(min_distance)
        c.ransac_threshold = float(1.0 - np.cos(np.deg2rad(5.0))) if thr is None else float(thr)
        c.seed, c.cos_a, c.sin_a = 0, float(cos_a), float(sin_a)
        if str(ransac_solver).upper() not in ("P3P", "GP3P"):
            raise ValueError("ransac_solver: P3P or GP3P")
        c.ransac_flags, c.reserved = (_lib.FLAG_GP3P if str(ransac_solver).upper() == "GP3P" else 0), 0
        self.cfg = c
        if getattr(model, "unwrap_table", None) is None:  # once per model
            model.unwrap_table = ctx.unwrap_prepare(model.omni_masks, model.map_x, model.map_y, (model.H, model.W))
        nbytes = ctx.sequence_workspace(c, self.W, self.slots)
        if nbytes <= 0:
            raise ValueError("bad sequence configuration")
        dev = ctx.device
        self.workspace = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        self._omni = [torch.zeros((self.W, model.H, model.W, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
        self._host = [torch.zeros((self.W, model.H, model.W, 3), dtype=torch.uint8).pin_memory() for _ in range(2)]
        self._host_np = [h.numpy() for h in self._host]
        self.spec = torch.zeros((self.W, 16), dtype=torch.float64, device=dev)
        self.spec2 = torch.zeros((self.W, 16), dtype=torch.float64, device=dev)
        self.one = torch.zeros((1, 16), dtype=torch.float64, device=dev)

    def _stage_host(self, images, buf):
        _stage_rows(self._host_np[buf], images)

    def _upload(self, buf, n):
        self._omni[buf][:n].copy_(self._host[buf][:n], non_blocking=True)

    def _front_end(self, n, first, buf):
        m = self.model
        self.ctx.sequence_front_end(self.rig, self.cfg, self.W, self.slots, self._omni[buf][:n], first, m.unwrap_table,
                                    m.mask_bits, m.pattern, self.workspace)

    def _track(self, ref_slots, cur_slots, seed, out):
        self.ctx.sequence_track(self.rig, self.cfg, self.W, self.slots, ref_slots, cur_slots, seed, self.workspace, out)

    def _counts(self, first, n):
        return self.ctx.sequence_frame_counts(self.cfg, self.W, self.slots, first, n, self.workspace)

    def _copy(self, src, dst):
        self.ctx.sequence_copy_slot(self.cfg, self.W, self.slots, src, dst, self.workspace)


def rgbd_solver_flags(pose_est_algorithm):
    """TrackerSE3.pose_est_algorithm (pose_est_tools.py:697; the names pyopengv.absolute_pose_ransac accepts, :89-107)
    -> the SOSVO_FLAG_* bits of the central RANSAC.  Unknown names raise, so that a batched engine can never silently run
    a different solver from the mirror tracker."""
    from . import _lib
    name = str(pose_est_algorithm).upper()
    table = {"EPNP": _lib.FLAG_EPNP, "KNEIP": 0, "GAO": _lib.FLAG_GP3P, "GP3P": _lib.FLAG_GP3P, "TWOPT": _lib.FLAG_TWOPT}
    if name not in table:
        raise ValueError("unknown pose_est_algorithm %r (have: %s)" % (pose_est_algorithm, ", ".join(sorted(table))))
    return table[name]


class RGBDCamConfig(object):
    """What the RGB-D hot path needs to know about the camera (host numbers): RGBDCamModel
    (omnistereo/camera_models.py:756-779), the RGBDFrame ranges (pose_est_tools.py:428-430, depth units) and
    TrackerRGBDSE3.bootstrap_tracker's |du| gate (pose_est_tools.py:956-958: 0.5 * 2 * center_x)."""

    def __init__(self, fx=525.0, fy=525.0, center_x=319.5, center_y=239.5, focal_length_m=1.0 / 1000.0, depth_is_Z=True,
                 min_range=0.8, max_range=7.0, f2f_max_hdiff=None, pct_good_matches=1.0):
        self.fx, self.fy, self.center_x, self.center_y = float(fx), float(fy), float(center_x), float(center_y)
        self.focal_length_m, self.depth_is_Z = float(focal_length_m), bool(depth_is_Z)
        self.min_range, self.max_range = float(min_range), float(max_range)
        self.f2f_max_hdiff = 0.5 * (2.0 * self.center_x) if f2f_max_hdiff is None else float(f2f_max_hdiff)
        self.pct_good_matches = float(pct_good_matches)

    def as_struct(self):
        from . import _lib
        c = _lib.RgbdCam()
        c.fx, c.fy, c.cx, c.cy, c.focal_length_m = self.fx, self.fy, self.center_x, self.center_y, self.focal_length_m
        c.depth_is_Z, c.reserved, c.min_range, c.max_range = 1 if self.depth_is_Z else 0, 0, self.min_range, self.max_range
        return c


class RGBDFrontEnd(object):
    """RGBDFrame.establish_keypoints (pose_est_tools.py:600-623) for F frames at once: [median,] gray,
    goodFeaturesToTrack on the whole image (or RGBDFrame.mask), ORB descriptors, depth back-projection, NaN / range
    filter, bearings.  All buffers in HBM; four asynchronous C-ABI calls."""

    def __init__(self, ctx, cam, nframes, image_shape=(480, 640), num_of_features=1000, kp_cap=None, frame_cap=None,
                 median_win_size=0, quality=0.01, min_distance=5.0, edge=31, mask=None):
        from . import orb_pattern
        assert isinstance(ctx, Context)
        self.ctx, self.cam_cfg, self.cam = ctx, cam, cam.as_struct()
        self.F = int(nframes)
        self.rows, self.cols = int(image_shape[0]), int(image_shape[1])
        self.num_of_features, self.median_win_size = int(num_of_features), int(median_win_size)
        self.quality, self.min_distance, self.edge = float(quality), float(min_distance), int(edge)
        # > 1024 selects the detector's large-mask variant (whole-image masks), see sosvo_detect_gft
        self.kp_cap = int(kp_cap) if kp_cap else int(min(4096, max(1088, -(-self.num_of_features // 64) * 64)))
        self.frame_cap = int(frame_cap) if frame_cap else self.kp_cap
        dev, F = ctx.device, self.F
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)  # noqa: E731
        self.bgr = z((F, self.rows, self.cols, 3), torch.uint8)
        self.depth = z((F, self.rows, self.cols), torch.float32)
        self.gray = z((F, self.rows, self.cols), torch.uint8)
        mb = np.ones((1, self.rows, self.cols), np.uint32) if mask is None else \
            (np.asarray(mask) != 0).astype(np.uint32).reshape(1, self.rows, self.cols)
        self.mask_bits = torch.from_numpy(mb).to(dev)       # one mask (bit 0): RGBDFrame.mask or the whole image
        self.pattern = torch.from_numpy(orb_pattern.orb_pattern()).to(dev)
        self.cos_a, self.sin_a = orb_pattern.angle_cos_sin(orb_pattern.GFT_KEYPOINT_ANGLE)
        self.kp, self.n = z((F, self.kp_cap, 2), torch.float32), z((F,), torch.int32)
        self.status, self.desc = z((F,), torch.int32), z((F, self.kp_cap, 32), torch.uint8)
        Fc = self.frame_cap
        self.frames = dict(m=z((F, Fc, 2), torch.float32), d=z((F, Fc, 32), torch.uint8), X=z((F, Fc, 3), torch.float64),
                           b=z((F, Fc, 3), torch.float64), M=z((F,), torch.int32))

    def load_frames(self, bgr, depth):
        """bgr [F,rows,cols,3] u8, depth [F,rows,cols] f32 (the camera's depth units, 0 = no reading)."""
        dev = self.ctx.device
        self.bgr.copy_((torch.from_numpy(np.ascontiguousarray(bgr)) if isinstance(bgr, np.ndarray) else bgr).to(dev))
        self.depth.copy_((torch.from_numpy(np.ascontiguousarray(depth, dtype=np.float32))
                          if isinstance(depth, np.ndarray) else depth).to(dev))

    def run(self):
        c = self.ctx
        c.median_gray(self.bgr, self.median_win_size, gray=self.gray)                              # :528, :531
        c.detect_gft(self.gray, self.mask_bits, self.F, 1, self.kp_cap, quality=self.quality,
                     min_distance=self.min_distance, max_corners=self.num_of_features, kp=self.kp, n=self.n,
                     status=self.status)                                                            # :544
        c.describe_orb(self.gray, self.kp, self.n, 1, self.pattern, self.cos_a, self.sin_a, edge=self.edge,
                       desc=self.desc)                                                              # :553
        c.rgbd_assemble(self.cam, self.kp, self.desc, self.n, self.depth, self.frame_cap, out=self.frames)  # :609-623


class RGBDPairPipeline(object):
    """The RGB-D (perspective) counterpart of ImageFrontEnd + FramePairPipeline (BASELINE config 5): for B
    independent pairs of (BGR image, depth map) frames, RGBDFrontEnd and then TrackerRGBDSE3.track_frame
    (pose_est_tools.py:896-954: frame-to-frame matching, |du| gate, central RANSAC, LM).
    Pair i tracks frame 2i+1 against frame 2i.  Everything stays in HBM; one step = 10 asynchronous C-ABI calls."""

    def __init__(self, ctx, cam, n_pairs, image_shape=(480, 640), num_of_features=1000, kp_cap=None, frame_cap=None,
                 median_win_size=0, quality=0.01, min_distance=5.0, edge=31, thr=None, max_iter=2000, adaptive=False,
                 seed=0, lm_iter=30, mask=None, pose_est_algorithm="EPNP"):
        """pose_est_algorithm: TrackerSE3.pose_est_algorithm (pose_est_tools.py:697): "EPNP" = 6-point samples solved by
        EPnP; "KNEIP" = P3P on 3 points + a 4th for disambiguation; "GAO" / "GP3P" = the depth formulation through the
        generalised solver; "TWOPT" = translation from 2 points.  Anything else raises ValueError."""
        from . import _lib
        self.ctx, self.cam_cfg = ctx, cam
        self.solver_flags = rgbd_solver_flags(pose_est_algorithm)
        self.epnp = bool(self.solver_flags & _lib.FLAG_EPNP)
        self.B, self.F = int(n_pairs), 2 * int(n_pairs)
        fe = RGBDFrontEnd(ctx, cam, self.F, image_shape, num_of_features, kp_cap, frame_cap, median_win_size, quality,
                          min_distance, edge, mask)
        self.front_end, self.frames, self.kp_cap, self.frame_cap = fe, fe.frames, fe.kp_cap, fe.frame_cap
        self.n, self.status, self.kp, self.desc, self.gray = fe.n, fe.status, fe.kp, fe.desc, fe.gray
        self.thr = float(1.0 - np.cos(np.deg2rad(5.0))) if thr is None else float(thr)
        self.max_iter, self.adaptive, self.seed, self.lm_iter = int(max_iter), bool(adaptive), int(seed), int(lm_iter)
        dev, F, B, Fc = ctx.device, self.F, self.B, self.frame_cap
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)  # noqa: E731
        self.ref_frame = torch.arange(0, F, 2, dtype=torch.int32, device=dev)
        self.cur_frame = torch.arange(1, F, 2, dtype=torch.int32, device=dev)
        self.keys, self.order = z((B, Fc, 1), torch.uint32), z((B, Fc), torch.int32)
        self.corr = dict(f=z((B, Fc, 3), torch.float64), p=z((B, Fc, 3), torch.float64), q=z((B, Fc), torch.int32),
                         t=z((B, Fc), torch.int32), n=z((B,), torch.int32))
        self.ransac = dict(T=z((B, 3, 4), torch.float64), mask=z((B, Fc), torch.uint8), idx=z((B, Fc), torch.int32),
                           n_inliers=z((B,), torch.int32), info=z((B, 4), torch.int32))
        self.T = z((B, 3, 4), torch.float64)
        self.lm_cost, self.lm_iters = z((B,), torch.float64), z((B,), torch.int32)

    def load_frames(self, bgr, depth):
        self.front_end.load_frames(bgr, depth)

    def detect(self):
        self.front_end.run()

    def track(self):
        c, fr, cfg = self.ctx, self.frames, self.cam_cfg
        c.match_hamming(fr["d"], fr["d"], fr["M"], fr["M"], k=1, keys=self.keys, q_slot=self.cur_frame,
                        t_slot=self.ref_frame)
        c.sort_matches(self.keys, fr["M"], order=self.order, q_slot=self.cur_frame)
        c.f2f_assemble_central(fr, self.ref_frame, self.cur_frame, self.keys, self.order, self.frame_cap,
                               pct_good_matches=cfg.pct_good_matches, max_hdiff=cfg.f2f_max_hdiff, out=self.corr)
        co = self.corr
        from . import _lib
        c.ransac_abs_pose(co["f"], co["p"], co["n"], self.thr, self.max_iter, seed=self.seed, adaptive=self.adaptive,
                          out=self.ransac, epnp=self.epnp, gp3p=bool(self.solver_flags & _lib.FLAG_GP3P),
                          twopt=bool(self.solver_flags & _lib.FLAG_TWOPT))                          # :915 (central)
        self.T.copy_(self.ransac["T"])
        c.refine_abs_pose(co["f"], co["p"], co["n"], self.T, idx=self.ransac["idx"], m=self.ransac["n_inliers"],
                          max_lm_iter=self.lm_iter, cost=self.lm_cost, iters=self.lm_iters)          # :937

    def step(self):
        self.detect()
        self.track()
        return self.T

    def results(self):
        """[B,16] f64 rows as FramePairPipeline.results()."""
        out = torch.empty((self.B, 16), dtype=torch.float64, device=self.ctx.device)
        out[:, :12] = self.T.reshape(self.B, 12)
        out[:, 12] = self.ransac["n_inliers"].to(torch.float64)
        out[:, 13] = self.corr["n"].to(torch.float64)
        out[:, 14] = self.ransac["info"][:, 2].to(torch.float64)
        out[:, 15] = self.ransac["info"][:, 0].to(torch.float64)
        return out


class RGBDPairBatch(object):
    """RGBDPairPipeline behind ONE C-ABI call per step (sosvo_rgbd_pair_batch): what a non-Python host would bind for
    BASELINE config 5.  Same arguments and results as RGBDPairPipeline; the workspace is allocated once."""

    def __init__(self, ctx, cam, n_pairs, image_shape=(480, 640), num_of_features=1000, kp_cap=None, frame_cap=None,
                 median_win_size=0, quality=0.01, min_distance=5.0, edge=31, thr=None, max_iter=2000, adaptive=False,
                 seed=0, lm_iter=30, mask=None, pose_est_algorithm="EPNP"):
        from . import _lib, orb_pattern
        self.ctx, self.cam_cfg, self.cam = ctx, cam, cam.as_struct()
        self.B = int(n_pairs)
        rows, cols = int(image_shape[0]), int(image_shape[1])
        kp_cap = int(kp_cap) if kp_cap else int(min(4096, max(1088, -(-int(num_of_features) // 64) * 64)))
        cos_a, sin_a = orb_pattern.angle_cos_sin(orb_pattern.GFT_KEYPOINT_ANGLE)
        c = _lib.RgbdBatchCfg()
        c.n_pairs, c.rows, c.cols, c.kp_cap, c.frame_cap = self.B, rows, cols, kp_cap, int(frame_cap) if frame_cap else kp_cap
        c.median_ksize, c.max_corners, c.edge = int(median_win_size), int(num_of_features), int(edge)
        c.ransac_max_iter, c.ransac_adaptive, c.lm_max_iter = int(max_iter), 1 if adaptive else 0, int(lm_iter)
        c.flags = rgbd_solver_flags(pose_est_algorithm)
        c.quality, c.min_distance = float(quality), float(min_distance)
        c.ransac_threshold = float(1.0 - np.cos(np.deg2rad(5.0))) if thr is None else float(thr)
        c.pct_good_matches, c.f2f_max_hdiff = float(cam.pct_good_matches), float(cam.f2f_max_hdiff)
        c.seed, c.cos_a, c.sin_a = int(seed), float(cos_a), float(sin_a)
        self.cfg, self.thr = c, c.ransac_threshold
        dev = ctx.device
        mb = np.ones((1, rows, cols), np.uint32) if mask is None else (np.asarray(mask) != 0).astype(np.uint32).reshape(1, rows, cols)
        self.mask_bits = torch.from_numpy(mb).to(dev)
        self.pattern = torch.from_numpy(orb_pattern.orb_pattern()).to(dev)
        nbytes = ctx.rgbd_pair_batch_workspace(c)
        if nbytes <= 0:
            raise ValueError("bad batch configuration")
        self.workspace = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        self.bgr = torch.zeros((2 * self.B, rows, cols, 3), dtype=torch.uint8, device=dev)
        self.depth = torch.zeros((2 * self.B, rows, cols), dtype=torch.float32, device=dev)
        self.out = torch.zeros((self.B, 16), dtype=torch.float64, device=dev)

    def load_frames(self, bgr, depth):
        dev = self.ctx.device
        self.bgr.copy_((torch.from_numpy(np.ascontiguousarray(bgr)) if isinstance(bgr, np.ndarray) else bgr).to(dev))
        self.depth.copy_((torch.from_numpy(np.ascontiguousarray(depth, dtype=np.float32))
                          if isinstance(depth, np.ndarray) else depth).to(dev))

    def step(self):
        """-> results [B,16] f64 (asynchronous)."""
        return self.ctx.rgbd_pair_batch(self.cam, self.cfg, self.bgr, self.depth, self.mask_bits, self.pattern, self.workspace,
                                        results=self.out)

    def results(self):
        return self.out


class RGBDSequenceEngine(_SequenceBase):
    """Sequence mode of the RGB-D path (demo_vo_rgbd.py's loop: one RGBDFrame per image, pose_est_tools.py:1440-1446, tracked
    against the current keyframe, :896-954) on the sosvo_rgbd_sequence_* entry points; push_window() takes (bgr, depth)
    tuples.  Same store layout, seeds and speculation as SequenceEngine."""

    def __init__(self, ctx, cam, window=32, image_shape=(480, 640), num_of_features=1000, kp_cap=None, frame_cap=None,
                 median_win_size=0, quality=0.01, min_distance=5.0, edge=31, thr=None, max_iter=210, adaptive=True, lm_iter=30,
                 mask=None, pose_est_algorithm="EPNP"):
        from . import _lib, orb_pattern
        assert isinstance(ctx, Context)
        self.ctx, self.cam_cfg, self.cam = ctx, cam, cam.as_struct()
        self._init_windows(window)
        rows, cols = int(image_shape[0]), int(image_shape[1])
        kp_cap = int(kp_cap) if kp_cap else int(min(4096, max(1088, -(-int(num_of_features) // 64) * 64)))
        cos_a, sin_a = orb_pattern.angle_cos_sin(orb_pattern.GFT_KEYPOINT_ANGLE)
        c = _lib.RgbdBatchCfg()
        c.n_pairs, c.rows, c.cols, c.kp_cap, c.frame_cap = self.W, rows, cols, kp_cap, int(frame_cap) if frame_cap else kp_cap
        c.median_ksize, c.max_corners, c.edge = int(median_win_size), int(num_of_features), int(edge)
        c.ransac_max_iter, c.ransac_adaptive, c.lm_max_iter = int(max_iter), 1 if adaptive else 0, int(lm_iter)
        c.flags = rgbd_solver_flags(pose_est_algorithm)
        c.quality, c.min_distance = float(quality), float(min_distance)
        c.ransac_threshold = float(1.0 - np.cos(np.deg2rad(5.0))) if thr is None else float(thr)
        c.pct_good_matches, c.f2f_max_hdiff = float(cam.pct_good_matches), float(cam.f2f_max_hdiff)
        c.seed, c.cos_a, c.sin_a = 0, float(cos_a), float(sin_a)
        self.cfg = c
        dev = ctx.device
        mb = np.ones((1, rows, cols), np.uint32) if mask is None else (np.asarray(mask) != 0).astype(np.uint32).reshape(1, rows, cols)
        self.mask_bits = torch.from_numpy(mb).to(dev)
        self.pattern = torch.from_numpy(orb_pattern.orb_pattern()).to(dev)
        nbytes = ctx.rgbd_sequence_workspace(c, self.W, self.slots)
        if nbytes <= 0:
            raise ValueError("bad sequence configuration")
        self.workspace = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        self._bgr = [torch.zeros((self.W, rows, cols, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
        self._depth = [torch.zeros((self.W, rows, cols), dtype=torch.float32, device=dev) for _ in range(2)]
        self._host_bgr = [torch.zeros((self.W, rows, cols, 3), dtype=torch.uint8).pin_memory() for _ in range(2)]
        self._host_depth = [torch.zeros((self.W, rows, cols), dtype=torch.float32).pin_memory() for _ in range(2)]
        self._np_bgr, self._np_depth = [h.numpy() for h in self._host_bgr], [h.numpy() for h in self._host_depth]
        self.spec = torch.zeros((self.W, 16), dtype=torch.float64, device=dev)
        self.spec2 = torch.zeros((self.W, 16), dtype=torch.float64, device=dev)
        self.one = torch.zeros((1, 16), dtype=torch.float64, device=dev)

    def _stage_host(self, images, buf):
        _stage_rows(self._np_bgr[buf], [im[0] for im in images])
        _stage_rows(self._np_depth[buf], [np.asarray(im[1], dtype=np.float32) for im in images])

    def _upload(self, buf, n):
        self._bgr[buf][:n].copy_(self._host_bgr[buf][:n], non_blocking=True)
        self._depth[buf][:n].copy_(self._host_depth[buf][:n], non_blocking=True)

    def _front_end(self, n, first, buf):
        self.ctx.rgbd_sequence_front_end(self.cam, self.cfg, self.W, self.slots, self._bgr[buf][:n], self._depth[buf][:n], first,
                                         self.mask_bits, self.pattern, self.workspace)

    def _track(self, ref_slots, cur_slots, seed, out):
        self.ctx.rgbd_sequence_track(self.cfg, self.W, self.slots, ref_slots, cur_slots, seed, self.workspace, out)

    def _counts(self, first, n):
        return self.ctx.rgbd_sequence_frame_counts(self.cfg, self.W, self.slots, first, n, self.workspace)

    def _copy(self, src, dst):
        self.ctx.rgbd_sequence_copy_slot(self.cfg, self.W, self.slots, src, dst, self.workspace)
