"""Self-consistency of the CPU oracle: sliding-sum restatement vs brute force, known-answer cases, and
independent numpy/scipy restatements of the LR check and the speckle filter."""
import numpy as np
import pytest
from scipy import ndimage


def rand_pair(rng, h, w, shift=5, noise=3):
    base = rng.integers(0, 256, (h, w + 64), dtype=np.uint8)
    base = ndimage.uniform_filter(base.astype(np.float32), 3).astype(np.uint8)
    L = base[:, 32:32 + w]
    R = np.clip(base[:, 32 + shift:32 + shift + w].astype(int) + rng.integers(-noise, noise + 1, (h, w)), 0, 255).astype(np.uint8)
    return np.ascontiguousarray(L), np.ascontiguousarray(R)


@pytest.mark.parametrize("h,w,nd,wsz,mind,cap,tex,uniq", [
    (40, 64, 16, 5, 0, 31, 10, 15),
    (37, 71, 32, 9, 0, 31, 0, 0),
    (48, 80, 16, 15, 0, 15, 10, 10),
    (33, 70, 16, 7, -8, 31, 5, 10),
    (33, 70, 16, 7, 4, 63, 5, 10),
    (30, 90, 48, 11, -60, 31, 0, 5),   # rofs > 0 branch
    (45, 60, 32, 21, 0, 31, 10, 10),   # window wider than the left margin
])
def test_sliding_matches_bruteforce(oracle, h, w, nd, wsz, mind, cap, tex, uniq):
    rng = np.random.default_rng(h * 1000 + w)
    L, R = rand_pair(rng, h, w)
    pl, pr = oracle.prefilter_xsobel(L, cap), oracle.prefilter_xsobel(R, cap)
    p = oracle.make_params(nd, wsz, cap, mind, tex, uniq)
    w2 = wsz // 2
    for row0, row1 in ((w2, h - w2), (0, h), (3, h - 2)):
        d1, c1 = oracle.find_correspondence(pl, pr, p, row0, row1)
        d2, c2 = oracle.find_correspondence(pl, pr, p, row0, row1, brute=True)
        assert np.array_equal(d1, d2)
        valid = d1 != (mind - 1) * 16
        assert np.array_equal(c1[valid], c2[valid])


def test_stripe_independence(oracle):
    """OpenCV splits the rows into stripes per thread; the result must not depend on the split."""
    rng = np.random.default_rng(7)
    L, R = rand_pair(rng, 60, 96)
    pl, pr = oracle.prefilter_xsobel(L, 31), oracle.prefilter_xsobel(R, 31)
    p = oracle.make_params(32, 9, 31, 0, 10, 10)
    whole, _ = oracle.find_correspondence(pl, pr, p, 4, 56)
    parts = np.full_like(whole, -16)
    for a, b in ((4, 17), (17, 18), (18, 40), (40, 56)):
        d, _ = oracle.find_correspondence(pl, pr, p, a, b)
        parts[a:b] = d[a:b]
    assert np.array_equal(whole, parts)


def test_prefilter_borders_and_odd_height(oracle):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (9, 20), dtype=np.uint8)
    out = oracle.prefilter_xsobel(img, 31)
    assert (out[8] == 31).all()                       # odd H: last row is cap
    assert (out[:, 0] == 31).all() and (out[:, -1] == 31).all()
    i = img.astype(int)
    dx = np.zeros_like(i)
    dx[:, 1:-1] = i[:, 2:] - i[:, :-2]
    ref = np.clip(dx[1] + 2 * dx[0] + dx[1], -31, 31) + 31   # reflect-101 at the top
    assert np.array_equal(out[0, 1:-1], ref[1:-1])
    img8 = img[:8]
    out8 = oracle.prefilter_xsobel(img8, 31)
    ref = np.clip(dx[6] + 2 * dx[7] + dx[6], -31, 31) + 31   # even H: reflect-101 at the bottom
    assert np.array_equal(out8[7, 1:-1], ref[1:-1])
    ref = np.clip(dx[2] + 2 * dx[3] + dx[4], -31, 31) + 31
    assert np.array_equal(out8[3, 1:-1], ref[1:-1])


def test_constant_shift_gives_constant_disparity(oracle):
    rng = np.random.default_rng(11)
    for shift in (3, 9, 14):
        L, R = rand_pair(rng, 48, 128, shift=shift, noise=0)
        p = oracle.make_params(16, 9, 31, 0, 10, 15)
        d = oracle.compute(p, L, R)
        roi = d[4:-4, 15 + 4 + 2:-4 - 2]
        valid = roi[roi >= 0]
        assert valid.size > 0.8 * roi.size
        assert (np.abs(valid.astype(int) - shift * 16) <= 8).all()   # exact match => c = 0, |sub-pixel| <= 1/2 px


def test_textureless_is_filtered_and_tie_prefers_largest_disparity(oracle):
    flat = np.full((40, 64), 100, np.uint8)
    p = oracle.make_params(16, 9, 31, 0, 10, 15)
    assert (oracle.compute(p, flat, flat) == -16).all()
    # texture test off, uniqueness off: every SAD is 0 -> first index wins -> largest disparity, sub-pixel 0
    p = oracle.make_params(16, 9, 31, 0, 0, 0)
    d = oracle.compute(p, flat, flat)
    assert (d[4:-4, 19:-4] == 15 * 16).all()


def test_range_does_not_fit_and_status_codes(oracle):
    img = np.zeros((32, 40), np.uint8)
    p = oracle.make_params(48, 5, 31, 0, 10, 15)   # lofs = 47 >= width
    assert (oracle.compute(p, img, img) == -16).all()
    bad = [
        (dict(num_disparities=20), -7), (dict(num_disparities=0), -7), (dict(block_size=4), -6),
        (dict(block_size=33), -6), (dict(prefilter_cap=0), -5), (dict(prefilter_cap=64), -5),
        (dict(texture_threshold=-1), -8), (dict(uniqueness_ratio=-1), -9), (dict(prefilter_size=4), -4),
        (dict(prefilter_type=2), -3),
    ]
    for kw, code in bad:
        args = dict(num_disparities=16, block_size=9)
        args.update(kw)
        assert oracle.compute_status(oracle.make_params(**args), 40, 32) == code, kw


def np_validate(disp, cost, mind, nd, tol_px):
    """Independent numpy restatement of SURVEY.md Appendix A.5."""
    disp = disp.copy()
    h, w = disp.shape
    inv = (mind - 1) * 16
    for y in range(h):
        d2 = np.full(w, inv, np.int64)
        c2 = np.full(w, np.iinfo(np.int64).max, np.int64)
        row = disp[y].astype(np.int64)
        for x in range(max(mind + nd, 0), w + min(mind, 0)):
            d = row[x]
            if d == inv:
                continue
            x2 = x - ((d + 8) >> 4)
            if c2[x2] > cost[y, x]:
                c2[x2], d2[x2] = cost[y, x], d
        for x in range(max(mind + nd, 0), w + min(mind, 0)):
            d = row[x]
            if d == inv:
                continue
            bad = []
            for xx in (x - (d >> 4), x - ((d + 15) >> 4)):
                bad.append(0 <= xx < w and d2[xx] > inv and abs(d2[xx] - d) > tol_px * 16)
            if all(bad):
                disp[y, x] = inv
    return disp


def test_validate_disparity_matches_numpy(oracle):
    rng = np.random.default_rng(5)
    h, w, nd = 12, 90, 32
    disp = rng.integers(0, (nd - 1) * 16, (h, w)).astype(np.int16)
    disp[rng.random((h, w)) < 0.3] = -16
    cost = rng.integers(0, 50, (h, w)).astype(np.int32)       # many ties on purpose
    for tol in (0, 1, 2):
        got = oracle.validate_disparity(disp, cost, 0, nd, tol)
        assert np.array_equal(got, np_validate(disp, cost, 0, nd, tol))
    got = oracle.validate_disparity(disp - 4 * 16, cost, -4, nd, 1)
    d4 = (disp - 64).copy()
    assert np.array_equal(got, np_validate(d4, cost, -4, nd, 1))


def test_lr_claim_key_with_the_disparity_picks_cv_winner():
    """Design claim behind the engine's LR kernel: keying a right-view column's claim by
    (cost, disparity) selects the same pixel as cv's (cost, then lowest x), because among the claimants of one column
    x2 = x - round(d/16) a smaller x means a strictly smaller d. Checked on random rows with many cost ties, negative
    disparities included."""
    rng = np.random.default_rng(17)
    for mind in (0, -20, 7):
        for _ in range(40):
            w, nd = 120, 48
            d = (rng.integers(0, nd * 16, w) + mind * 16).astype(np.int64)
            cost = rng.integers(0, 6, w).astype(np.int64)
            valid = rng.random(w) < 0.8
            x = np.arange(w)
            x2 = x - ((d + 8) >> 4)
            ok = valid & (x2 >= 0) & (x2 < w)
            for t in np.unique(x2[ok]):
                c = np.flatnonzero(ok & (x2 == t))
                by_x = c[np.lexsort((x[c], cost[c]))][0]            # lowest cost, then lowest x (cv)
                key = (cost[c] << 16) | ((d[c] & 0xffff) ^ 0x8000)  # the engine's 32-bit claim
                by_key = c[np.argmin(key)]
                assert by_x == by_key
                assert ((int(key.min()) & 0xffff) - 0x8000) == d[by_x]   # and the key carries the winner's disparity


def np_speckles(img, new_val, max_size, max_diff):
    """Independent restatement of Appendix A.6 as plain connected components (graph edges between
    4-neighbours that are both valid and within max_diff), via scipy.sparse.csgraph."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components

    h, w = img.shape
    v = img.astype(np.int64)
    valid = v != new_val
    idx = np.arange(h * w).reshape(h, w)
    rows, cols = [], []
    m = valid[:, :-1] & valid[:, 1:] & (np.abs(v[:, :-1] - v[:, 1:]) <= max_diff)
    rows.append(idx[:, :-1][m]); cols.append(idx[:, 1:][m])
    m = valid[:-1] & valid[1:] & (np.abs(v[:-1] - v[1:]) <= max_diff)
    rows.append(idx[:-1][m]); cols.append(idx[1:][m])
    r, c = np.concatenate(rows), np.concatenate(cols)
    g = coo_matrix((np.ones(r.size), (r, c)), shape=(h * w, h * w))
    _, lab = connected_components(g, directed=False)
    sizes = np.bincount(lab)
    out = img.copy()
    kill = valid.ravel() & (sizes[lab] <= max_size)
    out.ravel()[kill] = new_val
    return out


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_filter_speckles_is_plain_connected_components(oracle, seed):
    rng = np.random.default_rng(seed)
    h, w = 50, 70
    img = (rng.integers(0, 6, (h, w)) * 40).astype(np.int16)
    img = ndimage.median_filter(img, 3).astype(np.int16)
    img[rng.random((h, w)) < 0.25] = -16
    for max_size, max_diff in ((5, 0), (20, 40), (50, 32), (1, 100)):
        got = oracle.filter_speckles(img, -16, max_size, max_diff)
        assert np.array_equal(got, np_speckles(img, -16, max_size, max_diff)), (max_size, max_diff)


def test_valid_roi(oracle):
    assert oracle.valid_roi((0, 0, 640, 480), (0, 0, 640, 480), 0, 64, 21) == (73, 10, 557, 460)
    assert oracle.valid_roi((0, 0, 40, 30), (0, 0, 40, 30), 0, 64, 21) == (0, 0, 0, 0)
    assert oracle.valid_roi((10, 5, 600, 400), (20, 8, 500, 300), 0, 16, 9) == (39, 12, 477, 292)


def test_batch_matches_single(oracle):
    rng = np.random.default_rng(9)
    Ls, Rs = zip(*[rand_pair(rng, 40, 72) for _ in range(3)])
    p = oracle.make_params(16, 9, 31, 0, 10, 10, 20, 16, 1)
    got = oracle.compute_batch(p, np.stack(Ls), np.stack(Rs), threads=2)
    for i in range(3):
        assert np.array_equal(got[i], oracle.compute(p, Ls[i], Rs[i]))


def test_map_consumers_match_numpy_restatement(oracle):
    """Oracle functions for SURVEY 8f rank 1 against an independent numpy emulation of the reference's expressions
    (Stereo.cpp:157-199, SensorData.cpp:50-58): float where the C++ is float, double where it is double."""
    rng = np.random.default_rng(1)
    d = rng.integers(-16, 2000, (50, 80)).astype(np.int16)
    local = [0, 0, 1, 0.1, -1, 0, 0, 0.2, 0, -1, 0, 0.3]
    for mk in (dict(), dict(cx_r=600.5, local=local)):
        m = oracle.make_model(**mk)
        assert np.array_equal(oracle.decimate(d, 4), d[::4, ::4][:12, :20])

        def proj(px, py, disp):
            c = np.float32(m.cx_r - m.cx_l)
            dc = np.float32(np.float32(disp) + c)
            wx = np.float32((m.Tx_l / m.fx_l - m.Tx_r / m.fx_r) / np.float64(dc))
            wy = np.float32((m.Tx_l / m.fy_l - m.Tx_r / m.fy_r) / np.float64(dc))
            p = np.array([np.float32((np.float64(px) - m.cx_l) * np.float64(wx)),
                          np.float32((np.float64(py) - m.cy_l) * np.float64(wy)), np.float32(m.fx_l * np.float64(wx))], np.float32)
            if m.has_local:
                t = np.array(list(m.local), np.float32).reshape(3, 4)
                p = np.array([np.float32(np.float32(np.float32(t[r, 0] * p[0]) + np.float32(t[r, 1] * p[1])) + np.float32(t[r, 2] * p[2])) + t[r, 3]
                              for r in range(3)], np.float32)
            return p

        full = oracle.reproject(d, 4, m)
        with np.errstate(all="ignore"):
            for y in range(0, 50, 7):
                for x in range(0, 80, 3):
                    dd = np.float32(d[y, x]) / np.float32(16)
                    if dd > 0:
                        e = proj(np.float32(4 * x), np.float32(4 * y), dd)
                        assert np.isfinite(e).all()
                        assert np.array_equal(e.view(np.uint32), full[y, x].view(np.uint32)), (x, y)
                    else:
                        assert np.isnan(full[y, x]).all()
        kp = np.array([[10.7, 7.2], [79.9, 49.9], [-1, 3], [5, 50]], np.float32)
        k3 = oracle.keypoints3d(d, kp, m, 0.0, 0.0)
        assert np.isnan(k3[2]).all() and np.isnan(k3[3]).all()


def _prefilter_norm_incremental(src, winsize, ftzero):
    """prefilterNorm as OpenCV writes it (incremental column sums in ushort, padded vsum, sliding row sum, lookup table):
    an independent, literal restatement to check the oracle's closed form against."""
    h, w = src.shape
    wsz2 = winsize // 2
    scale_g = winsize * winsize // 8
    scale_s = (1024 + scale_g) // (scale_g * 2)
    scale_g *= scale_s
    OFS = 256 * 5
    TABSZ = OFS * 2 + 256
    tab = [0 if x - OFS < -ftzero else (ftzero * 2 if x - OFS > ftzero else x - OFS + ftzero) for x in range(TABSZ)]
    s = src.astype(np.int64)
    pad = wsz2 + 1
    vsum = np.zeros(w + 2 * pad, np.int64)      # vsum[pad + x]
    vsum[pad:pad + w] = (s[0] * (wsz2 + 2)) & 0xFFFF
    for y in range(1, wsz2):
        vsum[pad:pad + w] = (vsum[pad:pad + w] + s[min(y, h - 1)]) & 0xFFFF
    dst = np.zeros((h, w), np.uint8)
    for y in range(h):
        top = s[max(y - wsz2 - 1, 0)]
        bottom = s[min(y + wsz2, h - 1)]
        prev, curr, nxt = s[max(y - 1, 0)], s[y], s[min(y + 1, h - 1)]
        vsum[pad:pad + w] = (vsum[pad:pad + w] + bottom - top) & 0xFFFF
        for x in range(wsz2 + 1):
            vsum[pad - x - 1] = vsum[pad]
            vsum[pad + w + x] = vsum[pad + w - 1]
        tot = int(vsum[pad]) * (wsz2 + 1) + int(vsum[pad + 1:pad + wsz2 + 1].sum())
        val = ((int(curr[0]) * 5 + int(curr[1]) + int(prev[0]) + int(nxt[0])) * scale_g - tot * scale_s) >> 10
        dst[y, 0] = tab[val + OFS]
        for x in range(1, w - 1):
            tot += int(vsum[pad + x + wsz2]) - int(vsum[pad + x - wsz2 - 1])
            val = ((int(curr[x]) * 4 + int(curr[x - 1]) + int(curr[x + 1]) + int(prev[x]) + int(nxt[x])) * scale_g - tot * scale_s) >> 10
            dst[y, x] = tab[val + OFS]
        x = w - 1
        tot += int(vsum[pad + x + wsz2]) - int(vsum[pad + x - wsz2 - 1])
        val = ((int(curr[x]) * 5 + int(curr[x - 1]) + int(prev[x]) + int(nxt[x])) * scale_g - tot * scale_s) >> 10
        dst[y, x] = tab[val + OFS]
    return dst


@pytest.mark.parametrize("h,w,win,cap", [(24, 31, 9, 31), (17, 40, 5, 63), (30, 22, 15, 10), (12, 45, 21, 31)])
def test_prefilter_norm_closed_form_equals_incremental_form(oracle, h, w, win, cap):
    """Both restatements come from the same recollection of OpenCV (parity unpinned); this pins them to each other: the
    oracle's windowed definition vs the running-sum code with its ushort column sums, padding and initial row weights.
    (The incremental form needs the image to be taller than the half window, as OpenCV's initialisation loop assumes.)"""
    rng = np.random.default_rng(h * w + win)
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    assert h > win // 2
    assert np.array_equal(oracle.prefilter_norm(img, win, cap), _prefilter_norm_incremental(img, win, cap))


@pytest.mark.parametrize("cfg", [(320, 96, 64, 15, 0, 10, 10), (301, 77, 48, 9, -8, 0, 0), (260, 71, 32, 21, 4, 25, 200), (200, 60, 16, 5, 0, 5, 0),
                                 (400, 64, 128, 11, 0, 15, 10), (180, 50, 16, 21, -20, 10, 10)])
def test_u16_vectorised_variant_equals_the_scalar_restatement(oracle, pkg, cfg):
    """oracle/sbm_oracle_simd.c (bench.py's second CPU figure: 16-bit sums, key minimum, counting uniqueness test) against the
    scalar restatement: final maps, pre-LR maps and the cost plane, bit for bit -- windows / disparity counts / minDisparity of
    either sign / uniqueness 0 and large / texture thresholds, widths and heights that are not multiples of anything."""
    from u96_slam_amd import synth

    W, H, nd, w, mind, uniq, tex = cfg
    L, R = synth.make_batch(5, 2, W, H, max(nd, 16))
    rng = np.random.default_rng(W)
    L[1] = rng.integers(0, 256, (H, W))          # an uncorrelated pair: ties, failed uniqueness, large sums
    p = oracle.make_params(nd, w, 31, mind, tex, uniq, 40, 24, 1)
    assert oracle.simd_ok(p)
    for i in range(2):
        st0, a = oracle.compute(p, L[i], R[i], stages=True)
        with oracle.simd():
            st1, b = oracle.compute(p, L[i], R[i], stages=True)
        assert st0 == st1 == 0
        for k in ("disp", "pre_lr", "cost"):
            assert np.array_equal(a[k], b[k]), (cfg, i, k, int((a[k] != b[k]).sum()))
    assert not oracle.simd_ok(oracle.make_params(64, 27, 63))     # 27 * 27 * 126 does not fit 16 bits: the scalar path stays
