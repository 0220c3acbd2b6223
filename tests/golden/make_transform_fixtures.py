"""Generates tests/golden/transforms.npz by IMPORTING the reference's omnistereo.transformations and
omnistereo.common_tools (both import as-is, no placeholders needed) in the build container:
    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_transform_fixtures.py
Inputs and expected outputs only.  Pins the host helpers of the VO loop: quaternion <->
matrix, TUM entries, concatenation / inversion, the RPE metrics of the keyframe policy, and the TUM pose-file
reader with zero_up_wrt_origin."""
import collections
import collections.abc
import os
import sys
import tempfile

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "transforms.npz")


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference")
    collections.Iterable = collections.abc.Iterable
    np.float, np.int = float, int
    import omnistereo.transformations as tr
    import omnistereo.common_tools as ct
    rng = np.random.default_rng(2024)
    n = 24
    T = np.zeros((n, 4, 4))
    for i in range(n):
        ax = rng.normal(size=3)
        ang = rng.uniform(-np.pi, np.pi) if i else 0.0
        T[i] = tr.rotation_matrix(ang, ax)
        T[i, :3, 3] = rng.normal(size=3) * 3.0
    T[1, :3, :3] = np.diag([1.0, -1.0, -1.0])           # 180 degrees about x: w = 0
    noisy = T.copy()
    noisy[:, :3, :3] += rng.normal(scale=1e-4, size=(n, 3, 3))   # slightly non-orthonormal (as after LM + float I/O)
    out = dict(T=T, T_noisy=noisy)
    out["quat"] = np.stack([tr.quaternion_from_matrix(m, isprecise=False) for m in T])
    out["quat_noisy"] = np.stack([tr.quaternion_from_matrix(m, isprecise=False) for m in noisy])
    out["quat_matrix"] = np.stack([tr.quaternion_matrix(q) for q in out["quat"]])
    out["trans"] = np.stack([tr.translation_from_matrix(m) for m in T])
    out["inv"] = np.stack([tr.inverse_matrix(m) for m in T])
    out["concat3"] = np.stack([tr.concatenate_matrices(T[i], T[(i + 1) % n], T[(i + 2) % n]) for i in range(n)])
    out["rpe_rot"] = np.array([tr.rpe_rotation_metric(m) for m in noisy])
    out["rpe_trans"] = np.array([tr.rpe_translation_metric(m) for m in noisy])
    tum = np.zeros((n, 8))
    tum[:, 0] = np.arange(n) * 0.1
    tum[:, 1:4] = out["trans"]
    tum[:, 4:7] = out["quat"][:, 1:]
    tum[:, 7] = out["quat"][:, 0]
    tum[5, 4:] *= 1.7                                    # non-unit quaternion in the file
    out["tum_rows"] = tum
    out["from_tum"] = np.stack([tr.transform44_from_TUM_entry(list(r), has_timestamp=True) for r in tum])
    out["from_tum_scaled"] = np.stack([tr.transform44_from_TUM_entry(list(r[1:]), 1000.0, has_timestamp=False) for r in tum])
    with tempfile.TemporaryDirectory() as d:
        fn = os.path.join(d, "gt_TUM.txt")
        with open(fn, "w") as fh:
            fh.write("# timestamp tx ty tz qx qy qz qw\n")
            for r in tum:
                fh.write(" ".join(repr(float(v)) for v in r) + "\n")
        for zero_up in (False, True):
            poses7, mats = ct.get_poses_from_file(poses_filename=fn, input_units="m", output_working_units="mm", indices=[],
                                                  pose_format="tum", zero_up_wrt_origin=zero_up)
            out["file_poses7_%d" % zero_up] = np.array(poses7, dtype=np.float64)
            out["file_mats_%d" % zero_up] = np.stack(mats)
    # POV-Ray layout (common_tools.py:636-705): "tx, ty, tz, rot_x, rot_y, rot_z" per line, angles in degrees,
    # static-frame x -> y -> z rotations (quaternion_from_euler(..., 'sxyz'))
    pov = np.zeros((n, 6))
    pov[:, :3] = rng.normal(size=(n, 3)) * 2.0
    pov[:, 3:] = rng.uniform(-180.0, 180.0, size=(n, 3))
    pov[0, 3:] = 0.0
    pov[2, 3:] = [90.0, -90.0, 180.0]
    out["pov_rows"] = pov
    out["euler_quat_sxyz"] = np.stack([tr.quaternion_from_euler(*np.deg2rad(r[3:]), "sxyz") for r in pov])
    with tempfile.TemporaryDirectory() as d:
        fn = os.path.join(d, "gt_povray.txt")
        with open(fn, "w") as fh:
            fh.write("# tx, ty, tz, rx, ry, rz\n")
            for r in pov:
                fh.write(", ".join(repr(float(v)) for v in r) + "\n")
        for zero_up in (False, True):
            poses7, mats = ct.get_poses_from_file(poses_filename=fn, input_units="cm", output_working_units="mm", indices=[],
                                                  pose_format="povray", zero_up_wrt_origin=zero_up)
            out["pov_poses7_%d" % zero_up] = np.array(poses7, dtype=np.float64)
            out["pov_mats_%d" % zero_up] = np.stack(mats)
    out["units_mm_m"] = np.array([ct.get_length_units_conversion_factor("mm", "m"), ct.get_length_units_conversion_factor("m", "mm"),
                                  ct.get_length_units_conversion_factor("cm", "m"), ct.get_length_units_conversion_factor("m", "m")])
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
