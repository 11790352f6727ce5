"""GPU parity of the PL GFTT minimum-eigenvalue map (u96-slam_amd/csrc/sbm_gftt.hip) against the CPU restatement of
gftt_sbl.v / gftt_box.v / gftt_eig.v / gftt_obuf.v. Both take the exact floor of the square root, so they must agree
bit for bit; against the reference's CORDIC core the stated tolerance is +-1 LSB (oracle/sbm_oracle.h)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_gftt_map_and_max_register(pkg, oracle, golden):
    import torch

    bm = pkg.StereoBM.create(64, 21)
    rng = np.random.default_rng(9)
    imgs = [golden["rect_l"], golden["rect_r"], rng.integers(0, 256, (480, 640), dtype=np.uint8)]
    sat = (rng.integers(0, 2, (480, 640)) * 255).astype(np.uint8)   # 0 / 255 noise: drives the 16-bit limiters and the radicand cap
    imgs.append(sat)
    batch = np.stack(imgs)
    eig, mx = bm.gftt_eig(torch.from_numpy(batch).to("cuda:0"))
    eig, mx = eig.cpu().numpy(), mx.cpu().numpy()
    for i, im in enumerate(imgs):
        ref, rmax = oracle.gftt_eig(im)
        assert np.array_equal(eig[i], ref.astype(np.int64)), (i, int((eig[i] != ref).sum()))
        assert int(mx[i]) == rmax
        assert (eig[i][:2] == 0).all() and (eig[i][-2:] == 0).all() and (eig[i][:, 0] == 0).all() and (eig[i][:, -1] == 0).all()
    assert eig[3].max() > 30000


@pytest.mark.parametrize("W,H", [(3, 5), (65, 17), (130, 33), (1023, 511),
                                 # the two-columns-per-lane kernel (W >= 8): widths around its 126-column strips, odd and even
                                 # (the last Sobel column is a lane's first or second column), heights around its 3-row trips
                                 (7, 9), (8, 7), (9, 8), (125, 12), (126, 20), (127, 21), (128, 9), (252, 12), (253, 40), (254, 11),
                                 (379, 6), (640, 65), (1000, 130)])
def test_gftt_odd_sizes(pkg, oracle, W, H):
    import torch

    bm = pkg.StereoBM.create(64, 21)
    rng = np.random.default_rng(W)
    im = rng.integers(0, 256, (2, H, W), dtype=np.uint8)
    eig, mx = bm.gftt_eig(torch.from_numpy(im).to("cuda:0"))
    for i in range(2):
        ref, rmax = oracle.gftt_eig(im[i])
        assert np.array_equal(eig[i].cpu().numpy(), ref.astype(np.int64)) and int(mx[i]) == rmax
