"""GPU parity: libfbsmi primitives (through the C ABI) vs the CPU oracle, bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SIZES = [1, 2, 3, 7, 10, 255, 256, 257, 1000, 4096, 65536, 65537]


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("op", ["exp", "log", "log1p", "erfinv", "sqrt"])
def test_math_spec_bit_exact(op, oracle, dev):
    from fbs_amd import ops
    rng = np.random.default_rng(0)
    n = 1 << 20
    if op == "exp":
        x = np.concatenate([rng.uniform(-100, 89, n), rng.normal(0, 1, n), [-np.inf, 0., -87.3, -87.31, 88.7, 89.]])
    elif op == "log":
        x = np.concatenate([np.exp(rng.uniform(-100, 88, n)), rng.uniform(0, 2, n), [0., 1., np.inf, 1e-42]])
    elif op == "log1p":
        x = np.concatenate([-rng.uniform(0, 1, n) ** 2, rng.uniform(-1e-6, 1e-6, n), [-1., 0.]])
    elif op == "erfinv":
        x = np.concatenate([rng.uniform(-1, 1, n), np.tanh(rng.normal(0, 3, n)), [-1., 1., 0., -0.99999994, 0.99999994]])
    else:
        x = np.concatenate([np.exp(rng.uniform(-80, 80, n)), [0., 1., 2., 3.]])
    x = x.astype(np.float32)
    got = _np(ops.math_map(op, torch.from_numpy(x).to(dev)))
    want = getattr(oracle, op)(x)
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))


def test_division_bit_exact(oracle, dev):
    from fbs_amd import ops
    rng = np.random.default_rng(1)
    x = rng.normal(0, 1, 1 << 20).astype(np.float32)
    y = np.exp(rng.uniform(-20, 20, 1 << 20)).astype(np.float32)
    got = _np(ops.math_map("div", torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)))
    np.testing.assert_array_equal(got.view(np.uint32), oracle.div(x, y).view(np.uint32))


@pytest.mark.parametrize("n", SIZES)
def test_prng_bit_exact(n, oracle, dev):
    from fbs_amd import ops
    key = oracle.PRNGKey(1234 + n)
    np.testing.assert_array_equal(_np(ops.random_bits(key, (n,), device=dev)).view(np.uint32), oracle.random_bits(key, n))
    np.testing.assert_array_equal(_np(ops.uniform(key, (n,), device=dev)).view(np.uint32),
                                  oracle.uniform(key, (n,)).view(np.uint32))
    np.testing.assert_array_equal(_np(ops.normal(key, (n,), device=dev)).view(np.uint32),
                                  oracle.normal(key, (n,)).view(np.uint32))
    np.testing.assert_array_equal(_np(ops.randint(key, (n,), 0, 1000, device=dev)), oracle.randint(key, (n,), 0, 1000))
    np.testing.assert_array_equal(_np(ops.randint(key, (n,), -5, 7, device=dev)), oracle.randint(key, (n,), -5, 7))


def test_split_matches_golden_and_oracle(oracle):
    import os
    from fbs_amd import ops
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "keys_slice.npz"))
    mine = ops.split(ops.PRNGKey(666), 1000)
    np.testing.assert_array_equal(mine[gold["rows"]], gold["keys"])
    np.testing.assert_array_equal(mine, oracle.split(oracle.PRNGKey(666), 1000))


@pytest.mark.parametrize("n", SIZES + [300000, 3000001])
def test_cumsum_sum_logsumexp_bit_exact(n, oracle, dev):
    from fbs_amd import ops
    rng = np.random.default_rng(n)
    x = rng.uniform(0, 1, n).astype(np.float32)
    x /= x.sum()
    xt = torch.from_numpy(x).to(dev)
    np.testing.assert_array_equal(_np(ops.cumsum(xt)).view(np.uint32), oracle.cumsum(x).view(np.uint32))
    assert np.float32(_np(ops.tree_sum(xt))).view(np.uint32) == np.float32(oracle.tree_sum(x)).view(np.uint32)
    lw = rng.normal(0, 3, n).astype(np.float32)
    lwt = torch.from_numpy(lw).to(dev)
    assert np.float32(_np(ops.logsumexp(lwt))).view(np.uint32) == np.float32(oracle.logsumexp(lw)).view(np.uint32)
    for log_space in (True, False):
        got = _np(ops.normalise(lwt, log_space=log_space))
        np.testing.assert_array_equal(got.view(np.uint32), oracle.normalise(lw, log_space).view(np.uint32))
    # fbsmi_normalise_ess: the same normalisation plus the step's log-normaliser increment and the ESS
    out, lse, ess = ops.normalise(lwt, log_space=True, return_lse=True, return_ess=True)
    np.testing.assert_array_equal(_np(out).view(np.uint32), oracle.normalise(lw, True).view(np.uint32))
    assert np.float32(_np(lse)).view(np.uint32) == np.float32(oracle.logsumexp(lw)).view(np.uint32)
    assert np.float32(_np(ess)).view(np.uint32) == np.float32(oracle.ess(lw)).view(np.uint32)


def test_searchsorted_bit_exact(oracle, dev):
    from fbs_amd import ops
    rng = np.random.default_rng(5)
    for n in (1, 2, 10, 1000, 65536):
        w = rng.uniform(0, 1, n).astype(np.float32)
        c = oracle.cumsum(w / w.sum())
        q = np.concatenate([rng.uniform(-0.1, 1.1, 2000), c[: min(n, 50)], [0., 1.]]).astype(np.float32)
        got = _np(ops.searchsorted(torch.from_numpy(c).to(dev), torch.from_numpy(q).to(dev)))
        np.testing.assert_array_equal(got, oracle.searchsorted(c, q))


def _weights(n, seed, peaked=False):
    rng = np.random.default_rng(seed)
    lw = rng.normal(0, 2.0 if peaked else 0.3, n).astype(np.float32)
    import oracle as O
    return O.normalise(lw, False)


@pytest.mark.parametrize("n", [1, 2, 10, 100, 1000, 65536, 100001])
@pytest.mark.parametrize("kind", ["stratified", "systematic", "multinomial", "killing"])
def test_unconditional_resamplers_bit_exact(kind, n, oracle, dev):
    from fbs_amd.samplers import resampling as R
    for seed, peaked in ((n, False), (n + 1, True)):
        w = _weights(n, seed, peaked)
        key = oracle.PRNGKey(99 + seed)
        got = _np(getattr(R, kind)(torch.from_numpy(w).to(dev), key))
        np.testing.assert_array_equal(got, getattr(oracle, kind)(w, key))


@pytest.mark.parametrize("n", [2, 10, 100, 1000, 65536])
@pytest.mark.parametrize("kind", ["multinomial", "killing"])
def test_conditional_resamplers_bit_exact(kind, n, oracle, dev):
    from fbs_amd.samplers.csmc import resamplings as CR
    rng = np.random.default_rng(n)
    for trial in range(4):
        w = _weights(n, 1000 * trial + n, peaked=bool(trial & 1))
        key = oracle.PRNGKey(7 + trial)
        i, j = int(rng.integers(0, n)), int(rng.integers(0, n))
        for conditional in (True, False):
            got = _np(getattr(CR, kind)(key, torch.from_numpy(w).to(dev), i, j, conditional))
            want = getattr(oracle, "cond_" + kind)(key, w, i, j, conditional)
            np.testing.assert_array_equal(got, want)
            if conditional:
                assert got[j] == i


def test_conditional_systematic(oracle, dev):
    from fbs_amd.samplers.csmc import resamplings as CR
    w = _weights(1000, 3)
    key = oracle.PRNGKey(5)
    got = _np(CR.systematic(key, torch.from_numpy(w).to(dev), 0, 0, False))
    np.testing.assert_array_equal(got, oracle.cond_systematic(key, w, 0, 0, False))
    with pytest.raises(NotImplementedError):
        CR.systematic(key, torch.from_numpy(w).to(dev), 1, 2, True)


@pytest.mark.parametrize("n", [1, 10, 1000, 65536])
def test_categorical_and_force_move_bit_exact(n, oracle, dev):
    from fbs_amd import ops
    from fbs_amd.samplers.gibbs import force_move
    rng = np.random.default_rng(n)
    for trial in range(6):
        w = _weights(n, 31 * trial + n, peaked=bool(trial & 1))
        wt = torch.from_numpy(w).to(dev)
        key = oracle.PRNGKey(trial)
        assert int(ops.categorical(key, wt).item()) == int(oracle.choice(key, w, ()))
        k = int(rng.integers(0, n))
        gi, ga = force_move(key, wt, k)
        wi, wa = oracle.force_move(key, w, k)
        assert int(gi.item()) == wi
        assert np.float32(ga.item()).view(np.uint32) == np.float32(wa).view(np.uint32)
    # degenerate weight vector: all mass on k (gibbs.py:203-205 branch)
    if n > 1:
        w = np.zeros(n, np.float32)
        w[n // 2] = 1.0
        gi, _ = force_move(oracle.PRNGKey(3), torch.from_numpy(w).to(dev), n // 2)
        assert int(gi.item()) == oracle.force_move(oracle.PRNGKey(3), w, n // 2)[0]


def test_gather_set_backtrace(oracle, dev):
    from fbs_amd import ops
    rng = np.random.default_rng(2)
    for d in (1, 3, 4, 28 * 28):
        src = rng.normal(size=(500, d)).astype(np.float32)
        idx = rng.integers(0, 500, 777).astype(np.int32)
        got = _np(ops.take_rows(torch.from_numpy(src).to(dev), torch.from_numpy(idx).to(dev)))
        np.testing.assert_array_equal(got, src[idx])
        out = _np(ops.set_row(torch.from_numpy(src).to(dev), 17, torch.from_numpy(src[3]).to(dev)))
        ref = src.copy()
        ref[17] = src[3]
        np.testing.assert_array_equal(out, ref)
    T, n = 50, 64
    As = rng.integers(0, n, (T, n)).astype(np.int32)
    B = 13
    got = _np(ops.backtrace(torch.from_numpy(As).to(dev), torch.tensor(B, dtype=torch.int32, device=dev)))
    want = np.zeros(T + 1, np.int32)
    want[T] = B
    for k in range(T, 0, -1):
        want[k - 1] = As[k - 1, want[k]]
    np.testing.assert_array_equal(got, want)


def test_no_cpu_fallback():
    from fbs_amd import ops
    with pytest.raises(RuntimeError):
        ops.cumsum(torch.ones(8))


def test_kernel_normal_equals_the_definition_on_all_2_23_arguments(oracle, dev):
    """normal_from_bits (fbs_amd/csrc/fbsmi_device.h), the branch-free form every kernel draws its noise with, against
    fbsmi_bits_to_normal (include/fbsmi_math.h) evaluated on the device and by the C oracle: jax.random.normal uses the
    top 23 bits of a random word, so these are ALL the values the sampler can ever draw."""
    from fbs_amd import ops
    m = torch.arange(1 << 23, dtype=torch.int64, device=dev)
    for low in (0, 0x1FF):                                         # the 9 discarded bits must not matter
        bits = ((m << 9) | low).to(torch.int32)                    # wraps to the same 32-bit pattern
        words = bits.view(torch.float32)
        a = ops.math_map("bits_to_normal", words)
        b = ops.math_map("bits_to_normal_kernel", words)
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    want = oracle.bits_to_normal((np.arange(1 << 23, dtype=np.uint64) << np.uint64(9)).astype(np.uint32))
    got = b.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.isfinite(got).all() and abs(float(got.mean())) < 1e-3 and abs(float(got.std()) - 1.0) < 1e-3


def test_kernel_division_equals_ieee_division(dev):
    """div_by (fbs_amd/csrc/fbsmi_device.h): the reciprocal-and-two-corrections quotient the log-density kernels use for
    (v - mean)^2 / sd^2, against float32 `/` on 2^26 operand pairs spread over the exponents it is used for (and, outside
    them, the fall-back to `/`)."""
    from fbs_amd import ops
    g = torch.Generator(device=dev).manual_seed(1)
    n = 1 << 26
    mant = lambda: torch.rand(n, device=dev, generator=g) + 1.0
    ea = torch.randint(-70, 71, (n,), device=dev, generator=g).float()
    eb = torch.randint(-40, 11, (n,), device=dev, generator=g).float()
    a = mant() * torch.exp2(ea)
    b = mant() * torch.exp2(eb)
    a[:1000] = 0.0
    a[1000:2000] = torch.finfo(torch.float32).tiny * 3
    want = ops.math_map("div", a, b)
    got = ops.math_map("div_kernel", a, b)
    assert torch.equal(want.view(torch.int32), got.view(torch.int32))
    assert torch.equal(want, a / b)
