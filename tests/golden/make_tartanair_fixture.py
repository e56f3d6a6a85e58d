"""Builds tests/golden/tartanair_p000.npz from the reference's OWN unit-test asset
(/root/reference/Scripts/UnitTest/assets/test_sequence/TartanAir2_abs_P000: ground-truth depth, optical flow and poses of
the first frames of a TartanAir-v2 sequence, SURVEY.md §8(c) "Fixtures that pin results").

Decoding follows DataLoader/Dataset/TartanAir.py: depth = RGBA PNG whose B,G,R,A bytes (cv2 order) are a little-endian
float32 (:225-236); flow = 16-bit 3-channel PNG, flow = (u16 - 32768) / 64 on cv2 channels 0,1 (= PNG B, G), channel 2
(= PNG R) is the validity mask, 0 = valid (:274-292); pose_lcam_front.txt = x y z qx qy qz qw, NED, one line per frame
(:452-454); intrinsics fx = fy = cx = cy = 320, baseline 0.25 (DataLoader/Dataset/TartanAir2.py:82-85).

Only a crop is stored (rows 160:480, cols 96:544 -> 320 x 448; cx, cy shift accordingly) to keep the fixture small; flow is
kept as the raw uint16 samples so nothing is rounded.  Run in the build container (needs /root/reference):
    python tests/golden/make_tartanair_fixture.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from png_min import read_png  # noqa: E402

ROOT = "/root/reference/Scripts/UnitTest/assets/test_sequence/TartanAir2_abs_P000/"
N_FRAMES = 4
OY, OX, H, W = 160, 96, 320, 448


def main():
    depth, flow16, mask = [], [], []
    for i in range(N_FRAMES):
        d = read_png(ROOT + f"depth_lcam_front/{i:06d}_lcam_front_depth.png")
        d = np.ascontiguousarray(d[:, :, [2, 1, 0, 3]]).view("<f4")[..., 0]
        depth.append(d[OY:OY + H, OX:OX + W].copy())
    for i in range(N_FRAMES - 1):
        f = read_png(ROOT + f"flow_lcam_front/{i:06d}_{i + 1:06d}_flow.png")
        flow16.append(np.stack([f[OY:OY + H, OX:OX + W, 2], f[OY:OY + H, OX:OX + W, 1]]))   # (u, v) = cv2 channels 0, 1
        mask.append(f[OY:OY + H, OX:OX + W, 0].astype(np.uint8))
    poses = np.loadtxt(ROOT + "pose_lcam_front.txt")[:N_FRAMES]
    out = os.path.join(HERE, "tartanair_p000.npz")
    np.savez_compressed(out, depth=np.stack(depth).astype(np.float32), flow_u16=np.stack(flow16).astype(np.uint16),
                        flow_mask=np.stack(mask), poses=poses.astype(np.float64),
                        K=np.array([320.0, 320.0, 320.0 - OX, 320.0 - OY]), baseline=np.array(0.25),
                        crop=np.array([OY, OX, H, W]))
    print(out, os.path.getsize(out) / 1e6, "MB")


if __name__ == "__main__":
    main()
