#!/usr/bin/env python3
"""Headline benchmark: frame-pairs/s of the SOS VO front end hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the WHOLE hot path (unwrap -> median -> gray -> detect -> describe -> stereo
match -> triangulate -> frame-to-frame match -> RANSAC -> LM) over one batch of B independent synthetic
frame pairs per GPU; the omni frames are resident in HBM when the timed region starts.  For N > 1 there is
one rank per GPU: either a launcher (torch.distributed.run) started them, or -- `python bench.py --gpus N`
on its own -- this program starts them itself as child processes before it touches the GPU.  Ranks shard the
pairs (rank r owns the global pairs [r*B, (r+1)*B); no data-path collective) and all-gather the per-pair
pose records (16 doubles each) over RCCL at the end of every step.  Rank 0 prints ONE JSON line.
BASELINE config 4 (512 pairs over 8 GPUs) is `python bench.py --gpus 8 --pairs-per-gpu 64`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md


def b_alg_c2(H, W, kpts):
    """compulsory HBM bytes per frame pair: two BGR frames in; match index + distance out for
    4 matchings; inlier mask; pose."""
    return 2 * H * W * 3 + 4 * kpts * 12 + 2 * kpts + 96


def csrc_hash():
    """Short content hash of the library's device sources (csrc/*.hip, *.h): what a profile summary is tied to."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "vo_single_camera_sos_amd", "csrc", "*.hip")) +
                       glob.glob(os.path.join(ROOT, "vo_single_camera_sos_amd", "csrc", "*.h"))):
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def profile_meta(csv_path, run_cfg):
    """The <tag>_meta.json written beside a profile summary by scripts/profile_bench.sh (hash of the device sources,
    detector, solver, panorama width, pairs per launch, streams) compared with THIS run: -> dict(profile_commit,
    profile_csrc_hash, stale (bool or "unknown"), differs)."""
    tag = os.path.basename(csv_path).split("_")[0]
    meta_path = os.path.join(os.path.dirname(csv_path), tag + "_meta.json")
    if not os.path.exists(meta_path):
        return {"profile_commit": None, "stale": "unknown (no metadata beside the summary)"}
    meta = json.load(open(meta_path))
    differs = sorted(k for k, v in run_cfg.items() if k in meta and meta[k] != v)
    if meta.get("csrc_hash") and meta["csrc_hash"] != csrc_hash():
        differs.append("device sources (csrc_hash)")
    return {"profile_commit": meta.get("commit"), "profile_csrc_hash": meta.get("csrc_hash"), "stale": bool(differs), "differs": differs}


RUN_CFG = {"detector": "GFT", "ransac_solver": "P3P", "pano_width": 1440}   # main() overwrites it with the run's own


def profile_summaries(suffix):
    """Committed profile summaries profiles/*/*<suffix>, best match for THIS run first: those whose <tag>_meta.json names the
    run's detector / solver / panorama width, then (by name, newest round last) the ones without metadata; summaries of
    another configuration (meta present and different) come last."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "*" + suffix)))
    def rank(path):
        tag = os.path.basename(path).split("_")[0]
        meta_path = os.path.join(os.path.dirname(path), tag + "_meta.json")
        if not os.path.exists(meta_path):
            return 1
        meta = json.load(open(meta_path))
        same = all(meta.get(k) == v for k, v in RUN_CFG.items())
        return 2 if same else 0
    return sorted(paths, key=lambda q: (rank(q), q))   # callers walk it in reverse: best match first


def pmc_traffic(kernel, pairs_per_launch, pmc_csv=None):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 --pmc summary of this same command
    (profiles/*/*_pmc_hbm_per_kernel.csv, written by scripts/summarize_pmc.py from separate FETCH_SIZE and
    WRITE_SIZE passes; 2 * FETCH_SIZE + WRITE_SIZE as MI355X_MICROARCH.md prescribes for gfx950), rescaled to this
    run's batch.  (None, None) when no summary names the kernel: PMC passes cannot share a run with the timing."""
    import csv
    import glob
    label = kernel.strip("()").split("<")[0]
    paths = [pmc_csv] if pmc_csv else profile_summaries("pmc_hbm_per_kernel.csv")
    for path in reversed(paths):
        if not os.path.exists(path):
            continue
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if row.get("label") == label and int(row.get("pairs_per_launch") or 0) > 0:
                    scale = pairs_per_launch / float(row["pairs_per_launch"])
                    return float(row["hbm_bytes_per_launch"]) * scale, os.path.relpath(path, ROOT)
    return None, None


def traffic_step(pairs_per_step, streams, b_alg_pair, pmc_csv=None):
    """HBM bytes of ONE step summed over all kernels of the newest committed PMC summary (per-kernel bytes per launch x
    launches per step; the summary's launches cover (dispatches of the dominant kernel / streams) steps), per pair, and
    its ratio to the algorithmic bytes.  None without a summary."""
    import csv
    import glob
    paths = [pmc_csv] if pmc_csv else profile_summaries("pmc_hbm_per_kernel.csv")
    for path in reversed(paths):
        if not os.path.exists(path):
            continue
        rows = list(csv.DictReader(open(path)))
        dom = [r for r in rows if r["label"] in ("unwrap_median_gray_kernel", "median_gray_kernel")]
        if not dom or not int(dom[0]["pairs_per_launch"] or 0):
            continue
        steps_profiled = max(int(dom[0]["dispatches_fetch_pass"]), int(dom[0]["dispatches_write_pass"])) / float(streams)
        pairs_profiled_step = float(dom[0]["pairs_per_launch"]) * streams
        total, per_kernel = 0.0, {}
        for r in rows:
            if not r["label"].endswith("_kernel"):
                continue  # (torch's fill / copy kernels of the set-up)
            disp = max(int(r["dispatches_fetch_pass"]), int(r["dispatches_write_pass"]))
            b = float(r["hbm_bytes_per_launch"]) * disp / steps_profiled / pairs_profiled_step
            per_kernel[r["label"]] = b
            total += b
        top = dict(sorted(per_kernel.items(), key=lambda kv: -kv[1])[:6])
        return {"bytes_per_pair": total, "algorithmic_bytes_per_pair": b_alg_pair, "ratio_to_algorithmic": total / b_alg_pair,
                "bytes_per_step": total * pairs_per_step, "top_kernels_bytes_per_pair": top, "source": os.path.relpath(path, ROOT)}
    return None


def other_configs_subrecords(timeout_s=420):
    """BASELINE configs 3 and 5 as sub-records of the ONE bench line, so that the driver times them too: a child process
    (fresh interpreter, scripts/bench_other_configs.py) after this process's own GPU work is done."""
    import subprocess
    out = {}
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "bench_other_configs.py"), "--frames", "192", "--pairs", "128"],
                           capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
        for line in r.stdout.splitlines():
            if not line.startswith("{"):
                continue
            d = json.loads(line)
            key = d["config"].lower() if d["config"].startswith("C3") else "c5_" + d.get("algorithm", "").lower()
            d.pop("config", None)
            out[key] = d
        if r.returncode != 0 and not out:
            out["error"] = (r.stderr or "")[-400:]
    except Exception as e:  # the headline number does not depend on these
        out["error"] = repr(e)
    return out


def sequence_rgbd_subrecord(bgr, depth, mirror_frames=32):
    """The RGB-D counterpart of `sequence`: run_VO on an in-memory RGB-D sequence (640x480, radial depth) in sequence mode
    (frame_window 32) and through the per-frame mirror classes, host clock around run_VO."""
    import contextlib
    import io
    import tempfile
    from vo_single_camera_sos_amd.omnistereo.camera_models import RGBDCamModel
    from vo_single_camera_sos_amd.omnistereo.pose_est_tools import run_VO
    N = bgr.shape[0]
    out = {"frames": N, "data": "synthetic (one room, random-walk trajectory)",
           "tracker": "reference settings: GFT 1000, EPnP RANSAC <= 210 iterations (adaptive), 5 deg threshold, LM"}
    texts = {}
    for label, window, n in (("frame_window_32", 32, N), ("mirror_per_frame", 0, min(N, mirror_frames))):
        cam = RGBDCamModel(fx=554.256258, fy=554.256258, center_x=319.5, center_y=239.5, scaling_factor=1. / 1000.0,
                           do_undistortion=False, depth_is_Z=False, focal_length_m=1. / 1000.0)

        def frames(n=n):
            for k in range(n):
                yield k, bgr[k], depth[k]
        with tempfile.TemporaryDirectory() as d, contextlib.redirect_stdout(io.StringIO()):
            if label == "frame_window_32":   # first-use costs outside the clock (the same window: the allocator keeps its blocks)
                run_VO(None, cam, results_path=d, _live_frames=lambda: frames(min(N, 34)), frame_window=32)
                torch.cuda.synchronize()
            dts = []
            for _ in range(7 if label == "frame_window_32" else 1):   # (host-clocked, ~0.03 s a run: the median of seven)
                t0 = time.perf_counter()
                r = run_VO(None, cam, results_path=d, _live_frames=frames, frame_window=window)
                torch.cuda.synchronize()
                dts.append(time.perf_counter() - t0)
            dt = sorted(dts)[len(dts) // 2]
            texts[label] = open(os.path.join(d, "estimated_frame_poses_TUM.txt")).read()
        out[label] = {"frames": n, "frames_per_s": n / dt, "ms_per_frame": 1e3 * dt / n, "runs_s": dts, "tracked": r["tracked"],
                      "keyframes": len(r["keyframe_ids"])}
        if "sequence_mode" in r:
            out[label]["serial_tracking_calls"] = r["sequence_mode"]["serial_tracking_calls"]
            out[label]["host_wall_s"] = dict(r["sequence_mode"]["stage_s"], total=dt)
    n_m = out["mirror_per_frame"]["frames"]
    a = np.loadtxt(io.StringIO(texts["frame_window_32"]))[:n_m]
    b = np.loadtxt(io.StringIO(texts["mirror_per_frame"]))
    out["mirror_max_abs_pose_difference"] = float(np.abs(a - b).max())
    out["value"], out["unit"] = out["frame_window_32"]["frames_per_s"], "frames/s"
    return out


def sequence_subrecord(seq_omni, seq_poses, pano_width, mirror_frames=32):
    """SEQUENCE MODE (SURVEY 8(e) caveat; the reference's VO loop, pose_est_tools.py:1416-1628): run_VO on ONE synthetic
    sequence held in host memory -- every frame's front end computed once, `frame_window` frames per batched pass, tracking
    against keyframes from the device-resident frame store -- timed end to end by the host clock (engine set-up, host-to-
    device copies of the frames and the keyframe policy on the host included), beside the same loop at frame_window 1 and the
    per-frame mirror path (StereoPanoramicFrame on host arrays, one set of stage calls per frame: what the reference's loop
    does call by call).  The reference-style per-phase averages are the ones run_VO itself logs (process time)."""
    import contextlib
    import io
    import re
    import tempfile
    from vo_single_camera_sos_amd.omnistereo import transformations as tr
    from vo_single_camera_sos_amd.omnistereo.gum import synthetic_gums
    from vo_single_camera_sos_amd.omnistereo.panorama import Panorama
    from vo_single_camera_sos_amd.omnistereo.pose_est_tools import run_VO
    N = seq_omni.shape[0]
    out = {"frames": N, "panorama_width": pano_width, "data": "synthetic (one room, random-walk trajectory, steps <= 40 mm / 2 deg)",
           "tracker": "reference settings: GFT 1000 per mask, GP3P RANSAC <= 210 iterations (adaptive), 5 deg threshold, LM"}
    texts = {}
    for label, window, n in (("frame_window_32", 32, N), ("frame_window_1", 1, N), ("mirror_per_frame", 0, min(N, mirror_frames))):
        gs = synthetic_gums()
        for m in (gs.top_model, gs.bot_model):
            m.panorama = Panorama(m, width=pano_width)
        gs.make_annulus_masks(seq_omni.shape[1:3])

        def frames(n=n):
            for k in range(n):
                yield k, seq_omni[k], None
        with tempfile.TemporaryDirectory() as d, contextlib.redirect_stdout(io.StringIO()):
            if label == "frame_window_32":   # library scratch, unwrap table, first-use costs: outside the clock (same window size)
                run_VO(None, gs, results_path=d, _live_frames=lambda: frames(min(N, 34)), frame_window=32)
                torch.cuda.synchronize()
            dts = []
            for _ in range(7 if label == "frame_window_32" else 1):   # (host-clocked, ~0.03 s a run: the median of seven)
                t0 = time.perf_counter()
                r = run_VO(None, gs, results_path=d, _live_frames=frames, frame_window=window)
                torch.cuda.synchronize()
                dts.append(time.perf_counter() - t0)
            dt = sorted(dts)[len(dts) // 2]
            texts[label] = open(os.path.join(d, "estimated_frame_poses_TUM.txt")).read()
        rec = {"frames": n, "frames_per_s": n / dt, "ms_per_frame": 1e3 * dt / n, "runs_s": dts, "tracked": r["tracked"],
               "keyframes": len(r["keyframe_ids"])}
        for key, pat in (("image_read_avg_s", "Image Read Avg Time"), ("frame_setup_avg_s", "Frame Setup Avg Time"),
                         ("frame_tracking_avg_s", "Frame Tracking Avg Time"), ("overall_frame_vo_avg_s", "Overall Frame VO Avg Time")):
            mt = re.search(pat + r": ([0-9.eE+-]+) seconds", r["message"])
            rec[key] = float(mt.group(1)) if mt else None
        if "sequence_mode" in r:
            rec.update(serial_tracking_calls=r["sequence_mode"]["serial_tracking_calls"], windows=r["sequence_mode"]["windows"],
                       host_wall_s=dict(r["sequence_mode"]["stage_s"], total=dt))
        # per-hop accuracy against the planted trajectory (relative pose of consecutive frames)
        rot, tra = [], []
        P = r["poses"]
        for k in range(1, len(P)):
            Tg = [np.identity(4), np.identity(4)]
            for j, q in enumerate((k - 1, k)):
                Tg[j][:3, :3], Tg[j][:3, 3] = seq_poses[q][0], seq_poses[q][1] * 1e-3
            E = tr.rpe(tr.concatenate_matrices(tr.inverse_matrix(Tg[0]), Tg[1]),
                       tr.concatenate_matrices(tr.inverse_matrix(P[k - 1][1]), P[k][1]))
            rot.append(np.degrees(tr.rpe_rotation_metric(E)))
            tra.append(1e3 * tr.rpe_translation_metric(E))
        rec.update(per_hop_rotation_error_deg_median=float(np.median(rot)), per_hop_translation_error_mm_median=float(np.median(tra)))
        out[label] = rec
    out["pose_file_identical_window_32_vs_1"] = texts["frame_window_32"] == texts["frame_window_1"]
    n_m = out["mirror_per_frame"]["frames"]
    a = np.loadtxt(io.StringIO(texts["frame_window_32"]))[:n_m]
    b = np.loadtxt(io.StringIO(texts["mirror_per_frame"]))
    out["mirror_max_abs_pose_difference"] = float(np.abs(a - b).max())
    out["value"], out["unit"] = out["frame_window_32"]["frames_per_s"], "frames/s"
    hw = out["frame_window_32"].get("host_wall_s")
    if hw:   # the windows' time on the stream (events around a window's front end + speculative tracking; the loop enqueues
        # window k + 1 under the host's work on window k, so the host's WAIT is shorter than this) + the serial tracking calls
        gpu_s = hw.get("gpu_windows") or hw["wait_and_readback"]
        out["gpu_ms_per_frame"] = 1e3 * (gpu_s + hw["serial_track"]) / out["frame_window_32"]["frames"]
    return out


def opencv_opengv_baseline(omni, model, rig_kw, args, n_pairs):
    """if cv2 / pyopengv happen to be importable on this box, time the reference's own third-party calls
    (written against their public API; no reference file is shipped) on a bounded sample and report that as the primary
    CPU baseline.  Neither exists in the build image, so this leg normally reports availability only."""
    info = {"cv2": False, "pyopengv": False}
    try:
        import cv2  # noqa: F401
        info["cv2"] = True
    except Exception as e:
        info["cv2_error"] = type(e).__name__
    try:
        import pyopengv  # noqa: F401
        info["pyopengv"] = True
    except Exception as e:
        info["pyopengv_error"] = type(e).__name__
    if not info["cv2"]:
        return info
    try:
        import cv2
        mx, my = model.map_x.cpu().numpy(), model.map_y.cpu().numpy()
        masks = (model.mask_bits_host != 0)
        orb = cv2.ORB_create(nfeatures=args.features_per_mask)
        bf = cv2.BFMatcher(cv2.NORM_HAMMING, crossCheck=False)
        t0 = time.perf_counter()
        for i in range(n_pairs):
            per_frame = []
            for fr in (omni[2 * i], omni[2 * i + 1]):
                views = []
                for v in range(2):
                    pano = cv2.remap(fr, mx[v], my[v], cv2.INTER_LINEAR, borderMode=cv2.BORDER_CONSTANT, borderValue=0)
                    gray = cv2.cvtColor(cv2.medianBlur(pano, 11), cv2.COLOR_BGR2GRAY)
                    kps, descs = [], []
                    for m in range(model.nmask):
                        mk = (((model.mask_bits_host[v] >> m) & 1) * 255).astype(np.uint8)
                        pts = cv2.goodFeaturesToTrack(gray, args.features_per_mask, 0.01, 5, mask=mk, useHarrisDetector=False)
                        kp = [cv2.KeyPoint(float(p[0][0]), float(p[0][1]), 31) for p in (pts if pts is not None else [])]
                        kp, d = orb.compute(gray, kp)
                        kps.append(kp)
                        descs.append(d)
                    views.append((kps, descs))
                for m in range(model.nmask):   # static stereo per bucket
                    if views[0][1][m] is not None and views[1][1][m] is not None and len(views[0][1][m]) and len(views[1][1][m]):
                        bf.match(views[1][1][m], views[0][1][m])
                per_frame.append(views)
            for v in range(2):                 # frame-to-frame per view
                a = [d for d in per_frame[1][v][1] if d is not None and len(d)]
                b = [d for d in per_frame[0][v][1] if d is not None and len(d)]
                if a and b:
                    bf.match(np.concatenate(a), np.concatenate(b))
        dt = time.perf_counter() - t0
        info.update({"value": n_pairs / dt, "unit": "frame-pairs/s", "cores": 1, "kind": "reference-libraries",
                     "sample": "%d frame pairs: cv2 remap / medianBlur / goodFeaturesToTrack / ORB.compute / BFMatcher "
                               "(image stages + matching; RANSAC through pyopengv only if importable), %.1f s" % (n_pairs, dt)})
    except Exception as e:
        info["error"] = repr(e)
    return info


def host_cores():
    """Cores this process may run on (cgroup / affinity aware where the platform tells)."""
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        return max(1, os.cpu_count() or 1)


MIX_KEYS = {"unwrap_median_gray_kernel": ["median_gray_kernel<11, 1>"], "median_gray_kernel": ["median_gray_kernel<11, 0>"],
            "ransac_score_kernel": ["ransac_score_kernel<1, 4>"], "ransac_select_kernel": ["ransac_select_kernel<1>"],
            "gft_select_kernel": ["gft_select_kernel<2048, 256>", "gft_select_kernel<4096, 256>"],
            "match_hamming_mfma_kernel": ["match_hamming_mfma_kernel<2, 1>", "match_hamming_mfma_kernel<1, 1>"]}
SIMD_GINST_NS = 1024 / 2.4   # issue cycles (normalised to 2.4 GHz, as scripts/isa_mix.py prices them) -> 1 / (ginst/s) helper


def newest_isa_mix():
    """(kernels dict, relative path, commit) of the newest committed profiles/*/*isa_mix.json."""
    import glob
    for path in reversed(sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "*isa_mix.json")))):
        doc = json.load(open(path))
        return doc["kernels"], os.path.relpath(path, ROOT), doc.get("commit")
    return {}, None, None


def mix_cycles(label, mix, default=4.4):
    """Issue cycles (at 2.4 GHz) per VALU wave-instruction of the kernel a profile label names, from its OWN static mix."""
    keys = MIX_KEYS.get(label, [label])
    vals = [mix[k]["issue_cycles_per_valu_inst"] for k in keys if k in mix]
    if not vals:
        vals = [v["issue_cycles_per_valu_inst"] for k, v in mix.items() if k.split("<")[0] == label]
    return (sum(vals) / len(vals), True) if vals else (default, False)


def valu_issue(kernel, pairs_per_launch, avg_launch_s):
    """VALU instruction issue rate of `kernel` against the SIMDs' issue limit: the figure that actually bounds the
    dominant kernel.  Instruction count per launch from the newest committed SQ-counter summary
    (profiles/*/*_sq_per_kernel.csv: rocprofv3 --pmc SQ_INSTS_VALU ... of this command at B = 64, one stream); issue
    cost per instruction from the kernel's static instruction mix (profiles/*/*isa_mix.json, scripts/isa_mix.py) priced
    with the per-opcode rates MEASURED on the MI355X (scripts/valu_clock.hip, profiles/round3/valu_clock.txt: 2.50 real
    shader cycles per wave-instruction per SIMD for the fast class at 2.2-2.35 GHz, 4.2-4.3 for the slow class at
    2.37-2.39 GHz; quoted here normalised to 2.4 GHz: 2.7 / 4.4).  None without the summaries."""
    import csv
    import glob
    label = kernel.strip("()").split("<")[0]
    mix, mix_src, mix_commit = newest_isa_mix()
    cyc, _ = mix_cycles(label, mix)
    for path in reversed(profile_summaries("sq_per_kernel.csv")):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if row.get("label") == label:
                    insts = float(row["valu_insts"]) * pairs_per_launch / 64.0  # the SQ pass ran 64 pairs per launch
                    limit = 1024 * 2.4e9 / cyc  # wave-instructions/s of 1024 SIMDs
                    guide = 1024 * 2.4e9 / 2.0  # MI355X_MICROARCH.md: a wave64 VALU instruction issues over 2 cycles
                    return {"kernel": kernel, "valu_wave_insts_per_launch": insts, "achieved_ginst_s": insts / avg_launch_s / 1e9,
                            "issue_cycles_per_inst": cyc, "issue_limit_ginst_s": limit / 1e9,
                            "frac": insts / avg_launch_s / limit,
                            "frac_vs_guide_2cyc": insts / avg_launch_s / guide,
                            "guide_limit_ginst_s": guide / 1e9,
                            "measured_real_cycles": {"fast_class": 2.50, "slow_class": 4.27, "fast_class_8_waves_per_simd": 2.25,
                                                     "shader_clock_GHz_under_load": [2.2, 2.4],
                                                     "source": "profiles/round3/valu_clock.txt"},
                            "source": os.path.relpath(path, ROOT), "mix_source": mix_src, "mix_commit": mix_commit,
                            "note": "instruction count from the SQ pass; duration of isolated launches of the kernel "
                                    "(roofline.isolated_launch_ms): in the timed region the launches share the SIMDs with "
                                    "the other streams' kernels -- the whole step is accounted in valu_issue_step.  frac: "
                                    "against the ceiling of the kernel's own instruction mix at the MEASURED per-class issue "
                                    "costs (2.50 / 4.27 real shader cycles, s_memtime, at the 4 waves per SIMD this kernel "
                                    "runs at); frac_vs_guide_2cyc: against one wave64 VALU instruction per 2 SIMD-cycles at "
                                    "2.4 GHz, the guide's figure -- the SIMD approaches it only with 8 resident waves (2.25) "
                                    "and the clock drops to 2.2-2.3 GHz under fast-class load"}
    return None


def valu_issue_step(pairs_per_step, ms_per_step):
    """The whole step against the SIMDs' VALU issue rate: VALU wave-instructions of ALL kernels of a step (newest SQ
    summary, scaled from its 64 pairs per launch), each kernel priced by ITS OWN static instruction mix (newest
    profiles/*/*isa_mix.json; 4.4 for a kernel the mix file does not know), over 1024 SIMDs = the time the step would
    take if every issue slot were used.  frac = that bound / the measured time per step; frac_vs_guide_2cyc prices every
    instruction at the guide's 2 cycles instead."""
    import csv
    import glob
    mix, mix_src, mix_commit = newest_isa_mix()
    for path in reversed(profile_summaries("sq_per_kernel.csv")):
        cyc, total, dom, per_kernel, unknown = 0.0, 0.0, 0.0, {}, []
        steps_profiled = None
        with open(path) as fh:
            rows = list(csv.DictReader(fh))
        for row in rows:
            if row["label"] in ("unwrap_median_gray_kernel", "median_gray_kernel"):
                steps_profiled = float(row["dispatches"])   # one launch of the dominant kernel per step in the one-stream SQ pass
        if not steps_profiled:
            continue
        for row in rows:
            if not row["label"].endswith("_kernel"):
                continue  # (torch's own fill / copy kernels of the set-up)
            per_step = float(row["valu_insts"]) * float(row["dispatches"]) / steps_profiled
            c, known = mix_cycles(row["label"], mix)
            if not known:
                unknown.append(row["label"])
            is_dom = row["label"] in ("unwrap_median_gray_kernel", "median_gray_kernel")
            cyc += per_step * c
            total += per_step
            dom += per_step if is_dom else 0.0
            per_kernel[row["label"]] = {"share_of_valu_insts": per_step, "issue_cycles_per_inst": c}
        if total > 0:
            scale = pairs_per_step / 64.0
            bound_ms = cyc * scale / (1024 * 2.4e9) * 1e3
            guide_ms = total * 2.0 * scale / (1024 * 2.4e9) * 1e3
            for v in per_kernel.values():
                v["share_of_valu_insts"] /= total
            top = dict(sorted(per_kernel.items(), key=lambda kv: -kv[1]["share_of_valu_insts"])[:8])
            return {"valu_wave_insts_per_step": total * scale, "dominant_kernel_share": dom / total,
                    "issue_bound_ms_per_step": bound_ms, "measured_ms_per_step": ms_per_step, "frac": bound_ms / ms_per_step,
                    "guide_2cyc_bound_ms_per_step": guide_ms, "frac_vs_guide_2cyc": guide_ms / ms_per_step,
                    "per_kernel": top, "kernels_without_mix": unknown,
                    "source": os.path.relpath(path, ROOT), "mix_source": mix_src, "mix_commit": mix_commit}
    return None


LINE_BUDGET = 4000   # bytes: the driver keeps an 8 KB tail of stdout and parses the LAST line (20.7 KB did not parse)


def _sig(x, digits=5):
    """Floats to `digits` significant digits (bytes of the line, not precision of the measurement: the detail file keeps all)."""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float("%.*g" % (digits, x))
    return x


def _pick(d, keys):
    return {k: _sig(d[k]) for k in keys if isinstance(d, dict) and k in d}


def _clip(s, n):
    s = str(s)
    return s if len(s) <= n else s[: n - 3] + "..."


def compose_headline(out, detail_path=None):
    """The ONE line the driver parses, from the full record `out` (which goes to `detail_path` on disk): the contract's
    keys, `roofline`, `cpu_baseline`, and one number per sub-record.  Never more than LINE_BUDGET bytes: if a field should
    ever push it over, the optional groups are dropped in a fixed order (tests/test_bench_line.py builds a worst case)."""
    cfg = out.get("config", {})
    line = {k: _sig(out.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                          "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["metric"] = _clip(line["metric"], 120)
    line["config"] = dict(_pick(cfg, ("pairs_per_gpu", "global_pairs_per_step", "keypoints_per_view_mean",
                                      "stereo_points_per_frame_mean", "correspondences_per_pair_mean", "inliers_per_pair_mean",
                                      "tracked_ok", "rotation_error_deg_median", "translation_error_mm_median",
                                      "keypoint_capacity_hit")),
                          workload=_clip(cfg.get("workload", ""), 420), parallelism=_clip(cfg.get("parallelism", ""), 80))
    rf = out.get("roofline") or {}
    line["roofline"] = _pick(rf, ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_basis", "traffic", "traffic_source",
                                  "traffic_stale", "avg_launch_ms", "min_launch_ms", "launches", "pairs_per_launch",
                                  "isolated_launch_ms", "achieved_isolated", "frac_isolated", "algorithmic_bytes_per_pair",
                                  "algorithmic_bytes_per_launch", "profile_events_in_timed_region"))
    if "kernel" in line["roofline"]:
        line["roofline"]["kernel"] = _clip(line["roofline"]["kernel"], 60)
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = dict(_pick(cb, ("value", "unit", "cores", "kind")), sample=_clip(cb.get("sample", ""), 160))
    ca = out.get("cpu_baseline_all_cores")
    if ca and "value" in ca:
        line["cpu_baseline_all_cores"] = _pick(ca, ("value", "cores", "kind"))
    extra = {}
    if out.get("valu_issue"):
        extra["valu_issue_frac"] = _sig(out["valu_issue"].get("frac"))
    if out.get("valu_issue_step"):
        extra["valu_issue_step_frac"] = _sig(out["valu_issue_step"].get("frac"))
    if out.get("traffic_step"):
        extra["traffic_step_bytes_per_pair"] = _sig(out["traffic_step"].get("bytes_per_pair"))
        extra["traffic_step_stale"] = out["traffic_step"].get("stale")
    line["bounds"] = extra
    # one number (and its unit's worth of context) per sub-record; errors as short strings
    sub = {}
    for key in ("pcie_inclusive", "c_abi_streams", "orb_detector", "orb_detector_median11", "gp3p", "seeded_pattern", "pano_1200", "c3",
                "c3_2880", "c5_epnp", "c5_kneip", "sequence", "sequence_rgbd"):
        d = out.get(key)
        if not isinstance(d, dict):
            continue
        if "error" in d and "value" not in d:
            sub[key] = {"error": _clip(d["error"], 80)}
            continue
        e = {"value": _sig(d.get("value"))}
        if d.get("unit") and d["unit"] != "frame-pairs/s":
            e["unit"] = d["unit"]
        r2 = d.get("roofline")
        if isinstance(r2, dict):
            e["frac"] = _sig(r2.get("frac"), 3)
            e["kernel"] = _clip(str(r2.get("kernel", "")).strip("()").split("<")[0], 32)
        for k in ("keypoints_per_view_mean", "keypoints_per_view", "inliers_per_pair_mean", "ratio_to_engine", "h2d_GBps", "same_results",
                  "gpu_ms_per_frame"):
            if k in d:
                e[k] = _sig(d[k], 4)
        sub[key] = e
    for key in ("sub_error", "error"):
        if key in out:
            sub[key] = _clip(out[key], 120)
    line["sub"] = sub
    if out.get("accuracy_threshold_0p5deg") and "rotation_error_deg_median" in out["accuracy_threshold_0p5deg"]:
        line["accuracy_0p5deg"] = _pick(out["accuracy_threshold_0p5deg"], ("rotation_error_deg_median", "translation_error_mm_median"))
    if out.get("reference_libraries"):
        line["reference_libraries"] = _pick(out["reference_libraries"], ("cv2", "pyopengv", "value"))
    line["parity"] = "K1-K10 (OpenCV/OpenGV arithmetic) parity unpinned; geometry pinned by reference-generated fixtures"
    line["detail"] = detail_path
    for drop in (None, "reference_libraries", "accuracy_0p5deg", "bounds", "cpu_baseline_all_cores", "sub", "parity"):
        if drop:
            line.pop(drop, None)
        text = json.dumps(line, separators=(",", ":"))
        if len(text) <= LINE_BUDGET:
            return text
    line["config"].pop("workload", None)
    return json.dumps(line, separators=(",", ":"))[:LINE_BUDGET * 2]


def emit(out, detail_out):
    """Write the full record to `detail_out` (and a copy under gpurun_out/ when that directory exists: it is what gpurun
    brings back), then print the compact line as the LAST line of stdout."""
    paths = [detail_out] if detail_out else []
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")) and detail_out:
        paths.append(os.path.join(ROOT, "gpurun_out", os.path.basename(detail_out)))
    written = None
    for path in paths:
        try:
            with open(path, "w") as fh:
                json.dump(out, fh, indent=1)
            written = written or os.path.relpath(path, ROOT)
        except OSError:
            pass
    sys.stdout.flush()
    print(compose_headline(out, written), flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs-per-gpu", type=int, default=768,
                    help="B: independent frame pairs per GPU per step (weak scaling: fixed per GPU; C4's 512 pairs "
                         "over 8 GPUs is --pairs-per-gpu 64)")
    ap.add_argument("--detector", default="GFT", choices=["GFT", "ORB"],
                    help="GFT: the reference's default detector (pose_est_tools.py:684), always with ORB descriptors; "
                         "ORB: FAST/Harris pyramid detector (finds few corners on an 11x11-median-blurred panorama)")
    ap.add_argument("--features-per-mask", type=int, default=1000,
                    help="detector budget per azimuthal mask (reference default 1000, pose_est_tools.py:862)")
    ap.add_argument("--median-win-size", type=int, default=11,
                    help="StereoPanoramicFrame.median_win_size (11); 0 = no median blur (the ORB detector then finds its quota)")
    ap.add_argument("--pano-width", type=int, default=1440,
                    help="panorama columns.  The reference default (camera_models.py:3107) is 1200 -> 1200 x 122, where "
                         "GFT's minDistance 5 and ORB.compute's 31-px border cap the count at ~1250 keypoints per view; "
                         "1440 -> 1440 x 146 gives the ~2000 keypoints per view BASELINE's metric is quoted on")
    ap.add_argument("--iters", type=int, default=2000, help="RANSAC iterations, fixed (C2: 2000)")
    ap.add_argument("--cpu-pairs", type=int, default=64, help="frame pairs timed on the host for cpu_baseline")
    ap.add_argument("--streams", type=int, default=3,
                    help="HIP streams per GPU the batch is split over (the median launches take turns, the latency-bound "
                         "stages of the other parts overlap them); 1 = one stream")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the flow on one GPU)")
    ap.add_argument("--render-workers", type=int, default=0,
                    help="host processes rendering the synthetic frames (0 = auto; use 1 under rocprofv3, whose preloaded "
                         "tool initialises the GPU before this program forks)")
    ap.add_argument("--cpu-cores", type=int, default=0, help="processes of the all-cores CPU baseline (0 = os.cpu_count())")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-h2d", action="store_true", help="skip the extra PCIe-inclusive measurement (N = 1 only)")
    ap.add_argument("--no-isolated", action="store_true",
                    help="skip the isolated launches of the dominant kernel after the timed region (profile passes: the "
                         "per-kernel averages then cover the timed region only)")
    ap.add_argument("--pmc-csv", default=None, help="per-kernel PMC summary for roofline.traffic (default: newest in profiles/)")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--frames-cache", default=None,
                    help=".npz the rendered frame pairs are kept in / taken from (one rank only; the rocprofv3 passes of "
                         "scripts/profile_bench.sh render once, outside the profiler)")
    ap.add_argument("--ransac-solver", default="P3P", choices=["P3P", "GP3P"],
                    help="P3P: the three solve points of a sample from one mirror (BASELINE config 2's \"P3P RANSAC\"); GP3P: "
                         "generalised P3P on samples across both mirrors (what the reference's non-central RANSAC uses)")
    ap.add_argument("--no-sub", action="store_true",
                    help="skip the sub-records (ORB-detector path, GP3P hypotheses, BASELINE configs 3 and 5)")
    ap.add_argument("--sub-steps", type=int, default=12, help="timed steps of each in-process sub-record")
    ap.add_argument("--orb-features-per-mask", type=int, default=230,
                    help="ORB_create(nfeatures) per azimuthal mask of the orb_detector sub-record (12 masks x 230: ~2000 per view)")
    ap.add_argument("--sequence-frames", type=int, default=256,
                    help="frames of the synthetic sequence of the `sequence` sub-record (run_VO in sequence mode); 0 = skip")
    ap.add_argument("--sequence-child", action="store_true",
                    help="(internal) render the sequences, run the two sequence sub-records in THIS fresh process and print them "
                         "as one JSON line: a process that has created the batch engines loses ~15 % on the host-bound VO loop")
    ap.add_argument("--detail-out", default=os.path.join(ROOT, "bench_detail.json"),
                    help="file the FULL record goes to (per-kernel tables, notes, sub-records); stdout carries the compact line")
    ap.add_argument("--dump-records", default=None,
                    help="rank 0 writes the last step's gathered [N*B,16] records (global pair order) to this .npy file")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start the N ranks ourselves, as CHILD
    processes through torch.distributed.run (one rank per GPU, rendezvous on 127.0.0.1), and return their exit code.
    Runs before this process has made any GPU call (importing torch does not initialise the GPU) and never replaces
    the running program."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def cpu_baseline(omni, im, rig_kw, thr, iters, seed, n_pairs):
    """The reference's control flow on the CPU oracle (kind 'port'), one host core."""
    import refflow
    rp = refflow.RigParams(**rig_kw)
    t0 = time.perf_counter()
    for i in range(n_pairs):
        ref = refflow.frame_from_image(rp, im, omni[2 * i])
        cur = refflow.frame_from_image(rp, im, omni[2 * i + 1])
        refflow.track_pair(rp, ref, cur, thr, iters, seed=seed + i)
    dt = time.perf_counter() - t0
    return n_pairs / dt, dt


def cpu_baseline_all_cores(omni, im_kw, rig_kw, thr, iters, seed, cores, pairs_per_core):
    """The same flow with one frame pair per host thread-equivalent: `cores` spawned processes (fresh interpreters
    that never touch the GPU), each running the C oracle over its own pairs (baseline (ii))."""
    import multiprocessing
    import refflow
    n = cores * pairs_per_core
    reps = -(-n // (omni.shape[0] // 2))
    pool_omni = np.concatenate([omni] * reps)[: 2 * n] if reps > 1 else omni[: 2 * n]
    jobs = [(rig_kw, im_kw, pool_omni[2 * c * pairs_per_core: 2 * (c + 1) * pairs_per_core], thr, iters, seed + c * pairs_per_core)
            for c in range(cores)]
    with multiprocessing.get_context("spawn").Pool(cores) as pool:
        pool.map(refflow.pairs_worker, [(rig_kw, im_kw, j[2][:2], thr, 10, seed) for j in jobs])  # start-up outside the clock
        t0 = time.perf_counter()
        pool.map(refflow.pairs_worker, jobs)
        dt = time.perf_counter() - t0
    return n / dt, dt, n


def sequence_child(args):
    """`bench.py --sequence-child`: the two sequence sub-records in a fresh process (sequences rendered by forked workers
    before the GPU is touched), printed as one JSON line."""
    from vo_single_camera_sos_amd import synthetic
    from vo_single_camera_sos_amd.omnistereo.gum import synthetic_gums
    from vo_single_camera_sos_amd.omnistereo.panorama import Panorama
    gs = synthetic_gums()
    for m in (gs.top_model, gs.bot_model):
        m.panorama = Panorama(m, width=args.pano_width)
    gs.make_annulus_masks((480, 640))
    workers = args.render_workers if args.render_workers > 0 else max(1, min(16, host_cores()))
    seq_omni, seq_poses = synthetic.make_sequence(gs, args.sequence_frames, seed=args.seed + 7, workers=workers)
    seq_bgr, seq_depth, _ = synthetic.make_rgbd_sequence(min(128, args.sequence_frames), seed=args.seed + 9, workers=workers)
    out = {}
    try:
        out["sequence"] = sequence_subrecord(seq_omni, seq_poses, args.pano_width)
    except Exception as e:  # the headline number does not depend on it
        out["sequence"] = {"error": repr(e)}
    try:
        out["sequence_rgbd"] = sequence_rgbd_subrecord(seq_bgr, seq_depth)
    except Exception as e:
        out["sequence_rgbd"] = {"error": repr(e)}
    print(json.dumps(out), flush=True)
    return 0


def child_bench_subrecord(args, extra, env=None, timeout_s=240):
    """A variant of the headline step as a sub-record, measured by THIS program in a fresh child process (`--no-sub`): an
    engine created late in a process that has already built several measures 5 - 8 % low (the ORB-detector path 45.8 k as the
    fifth engine of the parent, 49.5 k on its own), and these records are the ones compared from round to round."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--no-sub", "--no-cpu", "--no-h2d", "--no-isolated", "--steps", str(args.sub_steps),
           "--warmup", "2", "--pairs-per-gpu", str(args.pairs_per_gpu), "--streams", str(args.streams), "--iters", str(args.iters),
           "--seed", str(args.seed), "--pano-width", str(args.pano_width), "--render-workers", str(args.render_workers),
           "--detail-out", os.path.join("/tmp", "sosvo_bench_child_%d.json" % os.getpid())] + list(extra)
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT, env=dict(os.environ, **(env or {})))
        for line in reversed(r.stdout.splitlines()):
            if line.startswith("{"):
                d = json.loads(line)
                cfg = d.get("config", {})
                return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
                        "pairs_per_step": cfg.get("global_pairs_per_step"), "keypoints_per_view_mean": cfg.get("keypoints_per_view_mean"),
                        "tracked_ok": cfg.get("tracked_ok"), "inliers_per_pair_mean": cfg.get("inliers_per_pair_mean"),
                        "roofline": d.get("roofline"), "process": "child (fresh interpreter, python bench.py --no-sub ...)"}
        return {"error": (r.stderr or "no output")[-300:]}
    except Exception as e:
        return {"error": repr(e)}


def sequence_subrecords_child(args, timeout_s=300):
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--sequence-child", "--sequence-frames", str(args.sequence_frames),
           "--pano-width", str(args.pano_width), "--seed", str(args.seed), "--render-workers", str(args.render_workers)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
        for line in reversed(r.stdout.splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"sequence": {"error": (r.stderr or "no output")[-300:]}}
    except Exception as e:
        return {"sequence": {"error": repr(e)}}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    from vo_single_camera_sos_amd import synthetic
    from vo_single_camera_sos_amd.omnistereo.gum import synthetic_gums
    from vo_single_camera_sos_amd.omnistereo.panorama import Panorama
    from vo_single_camera_sos_amd.parallel import gather_records, max_over_ranks
    from vo_single_camera_sos_amd.pipeline import OverlappedFramePairs, RigConfig

    if args.sequence_child:
        sys.exit(sequence_child(args))
    B = args.pairs_per_gpu
    H, W = 480, 640
    RUN_CFG.update(detector=args.detector, ransac_solver=args.ransac_solver, pano_width=args.pano_width)
    gs = synthetic_gums()
    for m in (gs.top_model, gs.bot_model):
        m.panorama = Panorama(m, width=args.pano_width)
    gs.make_annulus_masks((H, W))
    pano = gs.top_model.panorama
    geo = (pano.cols, pano.rows, pano.pixel_size, pano.cyl_height_max)
    rig_kw = dict(pano_top=geo, pano_bot=geo, F_top=gs.top_model.F[:3, 0], F_bot=gs.bot_model.F[:3, 0],
                  min_range=500.0, max_range=7000.0, stereo_min_disp=1.0, stereo_max_hdiff=2.5,
                  f2f_max_hdiff=0.125 * 0.5 * pano.cols, pct_good_matches=1.0)
    # frames are rendered on the host (forked workers) BEFORE this process touches the GPU
    workers = args.render_workers if args.render_workers > 0 else max(1, min(16, host_cores() // max(1, world)))
    # rank r owns the global pairs [r*B, (r+1)*B) of ONE job of world*B pairs: pair g is rendered from seed + g and
    # its RANSAC samples from seed + g, so the gathered records equal those of a single process over the same pairs
    cache = args.frames_cache if (args.frames_cache and world == 1) else None
    if cache and os.path.exists(cache):   # (the profile passes render once: scripts/profile_bench.sh)
        z = np.load(cache)
        assert z["omni"].shape[0] >= 2 * B and int(z["seed"]) == args.seed and int(z["pano_width"]) == args.pano_width, "stale frames cache"
        omni = np.ascontiguousarray(z["omni"][:2 * B])
        poses = [(z["R"][i], z["t"][i]) for i in range(B)]
    else:
        omni, poses = synthetic.make_frame_pairs(gs, B, seed=args.seed, workers=workers, first=rank * B)
        if cache:
            np.savez(cache, omni=omni, R=np.stack([p_[0] for p_ in poses]), t=np.stack([p_[1] for p_ in poses]), seed=args.seed,
                     pano_width=args.pano_width)
    seq_omni = None
    want_sequence = world == 1 and not args.no_sub and args.sequence_frames > 0   # (a child process renders and runs them)
    dist = None
    # (SOSVO_BENCH_FORCE_DIST=1: a launcher-started single rank also goes through init_process_group + the RCCL
    # gather -- tests/test_gpu_bench_rccl.py rehearses the N > 1 code path on the one-GPU box that way)
    if world > 1 or (os.environ.get("SOSVO_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        local_rank = local_rank % max(1, torch.cuda.device_count())  # (rehearsals with more ranks than GPUs)
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":  # RCCL on ROCm
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.dist_backend, rank=rank, world_size=world)
    n_gpus = world

    torch.cuda.set_device(local_rank)
    eng = OverlappedFramePairs(local_rank, gs, (H, W), RigConfig(**rig_kw), B, n_streams=args.streams,
                               num_of_features=args.features_per_mask, kp_cap=512, frame_cap=2048, max_iter=args.iters,
                               adaptive=False, seed=args.seed + rank * B, detection_method=args.detector,
                               ransac_solver=args.ransac_solver, median_win_size=args.median_win_size)
    model, dev = eng.model, eng.device
    eng.load_frames(omni)
    gathered = torch.empty((n_gpus * B, 16), dtype=torch.float64, device=dev) if dist else None

    def step():
        eng.step()
        rec = eng.results()
        if dist:
            gather_records(rec, out=gathered)  # 16 doubles per pair: latency-bound, one flat RCCL all-gather
        eng.consumed()
        return rec

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    eng.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rec = step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof = eng.profile_read()
    eng.profile_enable(False)
    elapsed = max_over_ranks(elapsed, dev)

    # The dominant kernel on its own (nothing else on the chip): in the timed region its launches share the SIMDs with the
    # other streams' kernels, at the lowest wave priority, so their durations say how the step is scheduled, not how good
    # the kernel is.  A few isolated launches of one part's K1+K2+K3 give the kernel's own duration.
    # With SOSVO_HINT_SHARED_DEVICE (set on the parts of a multi-stream engine) the kernel keeps to three workgroups per CU;
    # `isolated_launch_ms` is the kernel as it runs alone (hint off), `isolated_launch_ms_shared_hint` as the step launches it.
    iso_ms, iso_pairs, iso_hint_ms = None, None, None
    if rank == 0 and not args.no_isolated:
        part = eng.parts[0]

        def isolated():
            torch.cuda.synchronize()
            evs = []
            with torch.cuda.stream(part.stream):
                for _ in range(6):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(part.stream)
                    part.fe.run_images()
                    e1.record(part.stream)
                    evs.append((e0, e1))
            torch.cuda.synchronize()
            return float(np.median([a.elapsed_time(b) for a, b in evs[1:]]))

        if eng.S > 1:
            iso_hint_ms = isolated()
            part.ctx.set_hint_shared_device(False)
        iso_ms = isolated()
        if eng.S > 1:
            part.ctx.set_hint_shared_device(True)
        iso_pairs = part.hi - part.lo

    # PCIe-inclusive rate (never `value`): the same K steps with the omni frames handed over in pinned HOST memory and
    # copied per step, double-buffered on copy streams (OverlappedFramePairs.step_from_host).  One rank only.
    pcie = None
    if world == 1 and not args.no_h2d:
        rec_resident = rec.clone()
        pinned = torch.from_numpy(omni).pin_memory()
        for _ in range(2):
            eng.step_from_host(pinned)
            eng.results()
            eng.consumed()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.step_from_host(pinned)
            rec_h = eng.results()
            eng.consumed()
        torch.cuda.synchronize()
        dt_h = time.perf_counter() - t0
        pcie = {"value": B * args.steps / dt_h, "unit": "frame-pairs/s", "ms_per_step": 1e3 * dt_h / args.steps,
                "host_bytes_per_pair": 2 * H * W * 3, "h2d_GBps": B * args.steps * 2 * H * W * 3 / dt_h / 1e9,
                "same_results": bool(torch.equal(rec_h, rec_resident)),
                "note": "omni frames copied from pinned host memory every step (copy streams, two device buffers per "
                        "part: the copy of step k+1 overlaps the kernels of step k)"}
        del pinned

    if rank == 0 and args.dump_records:
        np.save(args.dump_records, (gathered if dist else rec).cpu().numpy())
    if rank == 0:
        rec = rec.cpu().numpy()
        n_kp = np.concatenate([p.fe.n.cpu().numpy().reshape(2, -1, model.nmask).sum(-1) for p in eng.parts], axis=1)  # [view, frame]
        M = np.concatenate([p.pipe.frames["M"].cpu().numpy() for p in eng.parts])
        cap_hit = any(int(p.fe.n.max().item()) >= p.fe.kp_cap for p in eng.parts)
        per_kernel = {}
        for name, ms in prof:
            s = per_kernel.setdefault(name, [0, 0.0, float("inf")])
            s[0] += 1
            s[1] += ms
            s[2] = min(s[2], ms)
        dom = max(per_kernel.items(), key=lambda kv: kv[1][1])
        dom_avg_s = dom[1][1] / dom[1][0] / 1e3
        dom_pairs = B / float(eng.S)  # one launch of the dominant kernel covers one stream's share of the batch
        kpts = int(round(float(n_kp.mean())))
        b_alg = b_alg_c2(H, W, kpts)
        b_alg_launch = b_alg * dom_pairs
        achieved = b_alg_launch / dom_avg_s / 1e9
        traffic, traffic_src = pmc_traffic(dom[0], dom_pairs, args.pmc_csv)
        ok = rec[:, 14] == 0
        def pose_errors(r):
            rot, tra = [], []
            for i in range(B):
                T = r[i, :12].reshape(3, 4)
                dR = T[:, :3].T @ poses[i][0]
                rot.append(np.degrees(np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1))))
                tra.append(float(np.linalg.norm(T[:, 3] - poses[i][1])))   # model units: mm
            return np.array(rot), np.array(tra)
        rot_err, tra_err = pose_errors(rec)
        out = {
            "metric": "frame-pairs/sec (detect+match+triangulate+RANSAC) on 640x480 omni, 2000 kpts",
            "value": n_gpus * B * args.steps / elapsed,
            "unit": "frame-pairs/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/f32/f64",  # images + Hamming in u8 / bits, corner response in f32, geometry + RANSAC + LM in f64
            "data": "synthetic",
            "config": {"workload": "C2: 640x480 BGR omni frame pair -> 2 x %dx%d panoramas per frame, %s, "
                                   "%s detector (budget %d per azimuthal mask x %d masks) + ORB descriptors, "
                                   "%d bucket + 2 frame-to-frame BF Hamming matchings, midpoint triangulation, "
                                   "non-central %s RANSAC %d iterations fixed, LM"
                                   % (pano.cols, pano.rows, "11x11 median" if args.median_win_size == 11 else
                                      "median window %d" % args.median_win_size, args.detector, args.features_per_mask, model.nmask,
                                      2 * model.nmask, "P3P (one-mirror samples)" if args.ransac_solver == "P3P" else
                                      "GP3P (samples across both mirrors)", args.iters),
                       "keypoint_capacity_hit": bool(cap_hit),
                       "pairs_per_gpu": B, "global_pairs_per_step": n_gpus * B,
                       "parallelism": "pairs sharded over ranks, dp%d; %d HIP streams per GPU" % (n_gpus, eng.S),
                       "hip_hardware_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default"),   # (the package asks for 32)
                       "keypoints_per_view_mean": float(n_kp.mean()), "stereo_points_per_frame_mean": float(M.mean()),
                       "correspondences_per_pair_mean": float(rec[:, 13].mean()),
                       "inliers_per_pair_mean": float(rec[:, 12].mean()), "tracked_ok": int(ok.sum()),
                       "rotation_error_deg_median": float(np.median(rot_err)),
                       "translation_error_mm_median": float(np.median(tra_err)),
                       "ransac_threshold_deg": 5.0,
                       "accuracy_note": "pose error against the planted motion (<= 100 mm, <= 5 deg).  The 0.7 deg / 4 cm medians "
                                        "are the reference's 5-degree threshold (pose_est_tools.py:675-676): correspondences with "
                                        "degrees of back-projection error stay in the LM set; the same frames at 0.5 degrees give "
                                        "the `accuracy_threshold_0p5deg` figures (profiles/round3/accuracy_sos.json: by threshold, "
                                        "solver and range bin -- the error does not depend on the points' range)"},
            "roofline": {"bound": "hbm", "kernel": dom[0], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "frac_basis": "avg_launch_ms (HIP events in the timed region, chip shared with the other streams); "
                                       "frac_isolated: the kernel alone",
                         "avg_launch_ms": dom_avg_s * 1e3, "min_launch_ms": dom[1][2], "launches": dom[1][0],
                         "pairs_per_launch": dom_pairs,
                         "profile_events_in_timed_region": "on; un-profiled C entry: sub.c_abi_streams.ratio_to_engine",
                         "isolated_launch_ms": iso_ms, "isolated_launch_ms_shared_hint": iso_hint_ms,
                         "achieved_isolated": b_alg * iso_pairs / (iso_ms * 1e-3) / 1e9 if iso_ms else None,
                         "frac_isolated": b_alg * iso_pairs / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if iso_ms else None,
                         "algorithmic_bytes_per_pair": b_alg, "algorithmic_bytes_per_launch": b_alg_launch,
                         "note": "no contraction anywhere (no MFMA); the dominant kernel is VALU-bound integer/bit work, "
                                 "so the HBM fraction is small by construction (SURVEY 8d)"},
            "valu_issue": valu_issue(dom[0], iso_pairs, iso_ms * 1e-3) if iso_ms else valu_issue(dom[0], dom_pairs, dom_avg_s),
            "valu_issue_step": None,
            "streams": eng.S,
            "kernels_ms_per_step": {k: v[1] / args.steps for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1][1])},
            "kernel_ms_per_step_total": sum(v[1] for v in per_kernel.values()) / args.steps,
            # the same sum with every launch at its kernel's SHORTEST duration in the timed region (~ the kernel with the
            # chip to itself): what the step's kernels cost back to back; the difference to the line above is time launches
            # spend sharing SIMDs / waiting for slots next to the other streams' kernels, not work
            "kernel_ms_per_step_min_total": sum(v[2] * v[0] for v in per_kernel.values()) / args.steps,
        }
        if n_gpus == 1 and not args.no_cpu:
            import refflow
            from vo_single_camera_sos_amd import orb_pattern
            ca, sa = orb_pattern.angle_cos_sin(-1.0)
            im_kw = dict(map_x=model.map_x.cpu().numpy(), map_y=model.map_y.cpu().numpy(),
                         omni_masks=model.omni_masks.cpu().numpy(), mask_bits=model.mask_bits_host, nmask=model.nmask,
                         max_corners=args.features_per_mask, pattern=model.pattern_host, cos_a=ca, sin_a=sa,
                         method=args.detector, kp_cap=eng.parts[0].fe.kp_cap)
            im = refflow.ImageModel(**im_kw)
            n_cpu = min(args.cpu_pairs, B)
            v, dt = cpu_baseline(omni, im, rig_kw, eng.thr, args.iters, args.seed, n_cpu)
            out["cpu_baseline"] = {"value": v, "unit": "frame-pairs/s", "cores": 1, "kind": "port",
                                   "sample": "%d of the same frame pairs through the C oracle (oracle/*.c) driven by "
                                             "tests/refflow.py, %.1f s" % (n_cpu, dt)}
            cores = max(1, min(args.cpu_cores if args.cpu_cores > 0 else host_cores(), 32))
            if cores > 1:
                try:
                    va, dta, na = cpu_baseline_all_cores(omni, im_kw, rig_kw, eng.thr, args.iters, args.seed, cores, 16)
                    out["cpu_baseline_all_cores"] = {"value": va, "unit": "frame-pairs/s", "cores": cores, "kind": "port",
                                                     "sample": "%d frame pairs, 16 per process, %d spawned processes, %.1f s"
                                                               % (na, cores, dta)}
                except Exception as e:  # the one-core figure above stays the reported baseline
                    out["cpu_baseline_all_cores"] = {"error": repr(e)}
        run_cfg = {"detector": args.detector, "ransac_solver": args.ransac_solver, "pano_width": args.pano_width,
                   "pairs_per_launch": int(dom_pairs), "streams": eng.S}
        if out["valu_issue"]:
            out["valu_issue_step"] = valu_issue_step(B, out["ms_per_step"])
            for rec_ in (out["valu_issue"], out["valu_issue_step"]):
                if rec_:   # (the SQ pass always runs one stream of 64 pairs per launch: compare the rest)
                    rec_.update(profile_meta(os.path.join(ROOT, rec_["source"]),
                                             {k: v for k, v in run_cfg.items() if k not in ("pairs_per_launch", "streams")}))
        out["traffic_step"] = traffic_step(B, eng.S, b_alg, args.pmc_csv)
        if out["traffic_step"]:
            out["traffic_step"].update(profile_meta(os.path.join(ROOT, out["traffic_step"]["source"]), run_cfg))
        if traffic_src:
            out["roofline"].update({"traffic_" + k: v for k, v in profile_meta(os.path.join(ROOT, traffic_src), run_cfg).items()})
        if n_gpus == 1 and not args.no_sub:
            # ---- sub-records of the same line (driver-timed): other detector / solver on the SAME frames, in this process
            def sub_engine(gs_=None, rig_kw_=None, num_of_features=None, pmc_tag=None, **kw):
                e2 = OverlappedFramePairs(local_rank, gs_ or gs, (H, W), RigConfig(**(rig_kw_ or rig_kw)), B, n_streams=args.streams,
                                          num_of_features=num_of_features or args.features_per_mask, kp_cap=512, frame_cap=2048,
                                          max_iter=args.iters, adaptive=False, seed=args.seed, **kw)
                e2.load_frames(omni)
                for _ in range(2):
                    e2.step()
                    e2.results()
                    e2.consumed()
                torch.cuda.synchronize()
                e2.profile_enable(True)
                t1 = time.perf_counter()
                for _ in range(args.sub_steps):
                    e2.step()
                    r2 = e2.results()
                    e2.consumed()
                torch.cuda.synchronize()
                dt2 = time.perf_counter() - t1
                prof2 = e2.profile_read()
                e2.profile_enable(False)
                r2 = r2.cpu().numpy()
                nk = np.concatenate([p.fe.n.cpu().numpy().reshape(2, -1, model.nmask).sum(-1) for p in e2.parts], axis=1)
                rec2 = {"value": B * args.sub_steps / dt2, "unit": "frame-pairs/s", "ms_per_step": 1e3 * dt2 / args.sub_steps,
                        "steps": args.sub_steps, "pairs_per_step": B, "keypoints_per_view_mean": float(nk.mean()),
                        "tracked_ok": int((r2[:, 14] == 0).sum()), "inliers_per_pair_mean": float(r2[:, 12].mean())}
                # roofline of this configuration: its dominant kernel by the library's own HIP-event profile
                pk = {}
                for name, ms in prof2:
                    e_ = pk.setdefault(name, [0, 0.0, float("inf")])
                    e_[0] += 1
                    e_[1] += ms
                    e_[2] = min(e_[2], ms)
                if pk:
                    # dominant kernel and frac from UNCONTENDED durations: the parts' launches overlap on the chip, so summed
                    # and average durations are shared time; the shortest launch in the timed region is
                    # the kernel nearly on its own
                    d2 = max(pk.items(), key=lambda kv: kv[1][2] * kv[1][0])
                    avg2 = d2[1][1] / d2[1][0] / 1e3
                    min2 = d2[1][2] / 1e3
                    ppl2 = B / float(e2.S)
                    ba2 = b_alg_c2(H, W, int(round(float(nk.mean()))))
                    import glob as _glob
                    tr2, src2 = (None, None)
                    if pmc_tag:
                        cands = sorted(_glob.glob(os.path.join(ROOT, "profiles", "*", "*" + pmc_tag + "*_pmc_hbm_per_kernel.csv")))
                        if cands:
                            tr2, src2 = pmc_traffic(d2[0], ppl2, cands[-1])
                    rec2["roofline"] = {"bound": "hbm", "kernel": d2[0], "achieved": ba2 * ppl2 / min2 / 1e9, "peak": HBM_PEAK_GBS,
                                        "unit": "GB/s", "frac": ba2 * ppl2 / min2 / 1e9 / HBM_PEAK_GBS,
                                        "frac_basis": "min_launch_ms (shortest launch in the timed region; dominant = largest "
                                                      "min x launches)",
                                        "frac_avg_launch": ba2 * ppl2 / avg2 / 1e9 / HBM_PEAK_GBS, "traffic": tr2,
                                        "traffic_source": src2, "avg_launch_ms": avg2 * 1e3, "min_launch_ms": d2[1][2],
                                        "algorithmic_bytes_per_pair": ba2, "pairs_per_launch": ppl2}
                    rec2["kernels_ms_per_step"] = {k: v[1] / args.sub_steps for k, v in sorted(pk.items(), key=lambda kv: -kv[1][1])[:8]}
                e2.close()
                return rec2
            def c_abi_streams():
                """The SAME step through the C ABI's one-call entry (what a non-Python host binds): B pairs per call over the
                library's internal HIP streams, calls enqueued back to back (sosvo_frame_pair_batch_streams_enqueue, two
                alternating record buffers, one join at the end) and, beside it, with the join after every call."""
                from vo_single_camera_sos_amd.device import Context
                from vo_single_camera_sos_amd.pipeline import FramePairBatch
                c2 = Context(local_rank)
                fb = FramePairBatch(c2, model, RigConfig(**rig_kw), B, num_of_features=args.features_per_mask, kp_cap=512,
                                    frame_cap=2048, max_iter=args.iters, seed=args.seed + rank * B, n_streams=args.streams,
                                    ransac_solver=args.ransac_solver)
                fb.load_frames(omni)
                bufs = [fb.out, torch.zeros_like(fb.out)]
                res = {}
                for label, joined in (("enqueue_back_to_back", False), ("join_every_call", True)):
                    for k in range(2):
                        fb.step() if joined else fb.enqueue(bufs[k & 1])
                    fb.join()
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for k in range(args.steps):
                        fb.step() if joined else fb.enqueue(bufs[k & 1])
                    t_host = time.perf_counter() - t1     # the host's share: the calls return once everything is enqueued
                    fb.join()
                    torch.cuda.synchronize()
                    dt2 = time.perf_counter() - t1
                    last = fb.out if joined else bufs[(args.steps - 1) & 1]
                    res[label] = {"value": B * args.steps / dt2, "unit": "frame-pairs/s", "ms_per_step": 1e3 * dt2 / args.steps,
                                  "host_ms_in_calls_per_step": 1e3 * t_host / args.steps,
                                  "same_records_as_engine": bool(np.array_equal(last.cpu().numpy(), rec))}
                res["value"], res["unit"], res["steps"], res["streams"] = res["enqueue_back_to_back"]["value"], "frame-pairs/s", args.steps, args.streams
                res["ratio_to_engine"] = res["value"] / out["value"]
                c2.close()
                return res
            try:
                # the same frames with the RANSAC threshold at 0.5 degrees: how well the (unpinned) solvers recover the planted motion
                e3 = OverlappedFramePairs(local_rank, gs, (H, W), RigConfig(**rig_kw), B, n_streams=args.streams,
                                          num_of_features=args.features_per_mask, kp_cap=512, frame_cap=2048, max_iter=args.iters,
                                          adaptive=False, seed=args.seed, detection_method=args.detector,
                                          ransac_solver=args.ransac_solver, thr=float(1.0 - np.cos(np.deg2rad(0.5))),
                                          median_win_size=args.median_win_size)
                e3.load_frames(omni)
                e3.step()
                r3 = e3.results().cpu().numpy()
                e3.close()
                ro3, tr3 = pose_errors(r3)
                out["accuracy_threshold_0p5deg"] = {"rotation_error_deg_median": float(np.median(ro3)), "rotation_error_deg_max": float(ro3.max()),
                                                    "translation_error_mm_median": float(np.median(tr3)), "translation_error_mm_max": float(tr3.max()),
                                                    "inliers_per_pair_mean": float(r3[:, 12].mean()), "tracked_ok": int((r3[:, 14] == 0).sum())}
            except Exception as e:
                out["accuracy_threshold_0p5deg"] = {"error": repr(e)}
            try:
                out["c_abi_streams"] = c_abi_streams()
            except Exception as e:
                out["c_abi_streams"] = {"error": repr(e)}
            try:
                if os.environ.get("SOSVO_ORB_PATTERN", "opencv") != "seeded":
                    # the SAME step with the seeded BRIEF table rounds 1-3 ran (round 4's default is OpenCV's learned table, whose
                    # descriptors match 1.67 x as many points on these frames: everything behind the matcher carries more work
                    # per pair).  Same build, same frames: the headline of rounds 1-3 and this round's compare through this record.
                    out["seeded_pattern"] = child_bench_subrecord(args, ["--detector", args.detector, "--ransac-solver", args.ransac_solver,
                                                                         "--features-per-mask", str(args.features_per_mask),
                                                                         "--median-win-size", str(args.median_win_size)],
                                                                  env={"SOSVO_ORB_PATTERN": "seeded"})
                    out["seeded_pattern"]["note"] = "rounds 1-3's workload: seeded BRIEF table instead of OpenCV's bit_pattern_31_"
                if args.detector != "ORB":
                    # the ORB detector AT THE METRIC'S LOAD (~2000 keypoints per view): without the median blur (the RGB-D
                    # frames' setting, pose_est_tools.py:427) and with the per-mask quota that yields that count; on the
                    # 11x11-median-blurred panoramas (the SOS frames' setting) FAST finds ~150 corners per view
                    out["orb_detector"] = child_bench_subrecord(args, ["--detector", "ORB", "--ransac-solver", args.ransac_solver,
                                                                       "--median-win-size", "0", "--features-per-mask",
                                                                       str(args.orb_features_per_mask)])
                    out["orb_detector"]["setting"] = "ORB_create(%d).detect per mask + compute, median_win_size 0" % args.orb_features_per_mask
                    out["orb_detector_median11"] = sub_engine(detection_method="ORB", ransac_solver=args.ransac_solver)
                if args.ransac_solver != "GP3P":
                    out["gp3p"] = child_bench_subrecord(args, ["--detector", args.detector, "--ransac-solver", "GP3P", "--features-per-mask",
                                                               str(args.features_per_mask), "--median-win-size", str(args.median_win_size)])
                if args.pano_width != 1200:
                    # the reference's default panorama width (demo_vo_sos.py: 1200 columns -> 1200 x 122 panoramas, fewer
                    # keypoints per view than the 2000 BASELINE's metric names) on the same omni frames
                    gs12 = synthetic_gums()
                    for m in (gs12.top_model, gs12.bot_model):
                        m.panorama = Panorama(m, width=1200)
                    gs12.make_annulus_masks((H, W))
                    p12 = gs12.top_model.panorama
                    geo12 = (p12.cols, p12.rows, p12.pixel_size, p12.cyl_height_max)
                    rig12 = dict(rig_kw, pano_top=geo12, pano_bot=geo12, f2f_max_hdiff=0.125 * 0.5 * p12.cols)
                    out["pano_1200"] = sub_engine(gs_=gs12, rig_kw_=rig12, detection_method=args.detector,
                                                  ransac_solver=args.ransac_solver)
                    out["pano_1200"]["panorama"] = "%d x %d" % (p12.cols, p12.rows)
            except Exception as e:
                out["sub_error"] = repr(e)
        if pcie is not None:
            out["pcie_inclusive"] = pcie
        if n_gpus == 1 and not args.no_cpu:
            out["reference_libraries"] = opencv_opengv_baseline(omni, model, rig_kw, args, min(8, B))
        sub_other = n_gpus == 1 and not args.no_sub
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()
    if rank == 0:
        if sub_other and want_sequence:
            torch.cuda.synchronize()
            out.update(sequence_subrecords_child(args))
        if sub_other:
            torch.cuda.synchronize()
            out.update(other_configs_subrecords())
        emit(out, args.detail_out)
    return


if __name__ == "__main__":
    main()
