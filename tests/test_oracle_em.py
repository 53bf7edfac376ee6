"""CPU: the numpy restatement of the image closures (oracle/em.py) against independent float64 / scipy
references of the same formulas (experiments/imgs/inpainting.py:102-147) and against the C oracle's tree sum."""
import numpy as np
import scipy.stats


def test_tree_sum_rows_is_orc_sum(oracle):
    from oracle import em
    rng = np.random.default_rng(0)
    for d in (1, 2, 3, 7, 64, 255, 256, 257, 559, 1000):
        x = rng.normal(size=(5, d)).astype(np.float32)
        got = em.tree_sum_rows(x)
        want = np.array([oracle.tree_sum(r) for r in x], np.float32)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), d


def test_tables_concat_unpack_roundtrip(oracle):
    from oracle import em, images
    shape = (16, 16, 3)
    _, unobs, obs = images.gen_inpaint_mask(oracle.PRNGKey(1), shape, 8, 8)
    u_off, v_off, role = em.element_tables(unobs, obs, 3)
    assert sorted(np.concatenate([u_off, v_off]).tolist()) == list(range(16 * 16 * 3))
    rng = np.random.default_rng(1)
    us = rng.normal(size=(6, u_off.size)).astype(np.float32)
    vp = rng.normal(size=v_off.size).astype(np.float32)
    A = np.array([5, 0, 0, 3], np.int32)
    img = em.concat(us, A, vp, role)
    ref = images.concat(us[A].reshape(4, -1, 3), np.broadcast_to(vp.reshape(-1, 3), (4, v_off.size // 3, 3)), shape,
                        unobs, obs)
    assert np.array_equal(img, ref.reshape(4, -1))
    x, y = images.unpack(img.reshape(4, *shape), shape, unobs, obs)
    assert np.array_equal(x.reshape(4, -1), us[A]) and np.array_equal(y[2].reshape(-1), vp)
    h = em.to_bf16_bits(img)
    assert np.all(np.abs(em.from_bf16_bits(h) - img) <= np.abs(img) * 2.0 ** -8)


def test_finish_matches_float64_formulas(oracle):
    from oracle import em, images
    shape = (12, 12, 2)
    _, unobs, obs = images.gen_inpaint_mask(oracle.PRNGKey(2), shape, 5, 5)
    u_off, v_off, role = em.element_tables(unobs, obs, 2)
    rng = np.random.default_rng(2)
    n = 11
    us = rng.normal(size=(n, u_off.size)).astype(np.float32)
    net = rng.normal(size=(n, role.size)).astype(np.float32)
    v, vp = rng.normal(size=v_off.size).astype(np.float32), rng.normal(size=v_off.size).astype(np.float32)
    cx, cs, dt, sd = 0.4, 1.7, 0.01, 0.13
    key = oracle.PRNGKey(3)
    A = rng.integers(0, n, n).astype(np.int32)
    for mode in (0, 1):
        us_new, lw = em.finish(us, A, net, mode, cx, cs, dt, sd, v, vp, key, n, 0, 4, np.ones(u_off.size, np.float32),
                               u_off, v_off)
        x = us[A].astype(np.float64)
        z = oracle.normal(key, (n, u_off.size)).astype(np.float64)
        du_ = net[:, u_off] if mode else cx * x + cs * net[:, u_off]
        want = x + du_ * dt + sd * z
        want[4] = 1.0
        assert np.allclose(us_new, want, rtol=1e-5, atol=1e-6)
        dv_ = net[:, v_off] if mode else cx * vp[None].astype(np.float64) + cs * net[:, v_off]
        want_lw = scipy.stats.norm.logpdf(v[None].astype(np.float64), vp[None] + dv_ * dt, sd).sum(axis=1)
        assert np.allclose(lw, want_lw, rtol=1e-5)
        tl = em.transition_logpdf(us, net, mode, cx, cs, dt, sd, us[0], u_off)
        du2 = net[:, u_off] if mode else cx * us.astype(np.float64) + cs * net[:, u_off]
        want_tl = scipy.stats.norm.logpdf(us[0][None].astype(np.float64), us + du2 * dt, sd).sum(axis=1)
        assert np.allclose(tl, want_tl, rtol=1e-5)
    # a slice of the rows equals the slice of the whole
    whole, _ = em.finish(us, None, net, 0, cx, cs, dt, sd, v, vp, key, n, 0, -1, None, u_off, v_off, want_lw=False)
    part, _ = em.finish(us[3:8], None, net[3:8], 0, cx, cs, dt, sd, v, vp, key, n, 3, -1, None, u_off, v_off, want_lw=False)
    assert np.array_equal(whole[3:8], part)
