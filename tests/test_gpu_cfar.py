"""GPU parity: CFAR through the C ABI (sonar_slam_amd.cfar shim) vs the CPU oracle, bit-exact."""
import ctypes as C

import numpy as np
import pytest

import oracle
from sonar_slam_amd import _lib, cfar, synth

pytestmark = pytest.mark.gpu

ALGS = ["CA", "SOCA", "GOCA", "OS"]
FN = {"CA": cfar.ca, "SOCA": cfar.soca, "GOCA": cfar.goca, "OS": cfar.os}
FN2 = {"CA": cfar.ca2, "SOCA": cfar.soca2, "GOCA": cfar.goca2, "OS": cfar.os2}


def _args(shipped_cfar, alg):
    return shipped_cfar.params[alg]


def _oracle(img, alg, p, thr=False):
    k = p[2] if alg == "OS" else 0
    return oracle.cfar(img, alg, p[0], p[1], p[-1], k=k, want_threshold=thr)


@pytest.mark.parametrize("alg", ALGS)
@pytest.mark.parametrize("shape", [(1024, 512), (300, 256), (130, 260), (120, 37), (60, 512), (51, 8),
                                   (50, 8), (7, 5), (1, 1)])
def test_mask_bit_exact_random(alg, shape, shipped_cfar):
    rng = np.random.default_rng(abs(hash((alg, shape))) % 2**32)
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    p = _args(shipped_cfar, alg)
    got = FN[alg](img, *p)
    assert got.dtype == np.uint8 and got.shape == img.shape
    assert np.array_equal(got, _oracle(img, alg, p))


@pytest.mark.parametrize("alg", ["CA", "SOCA", "GOCA"])
def test_ring_and_generic_kernels_agree_with_oracle(alg, shipped_cfar, ctx):
    img = synth.sonar_frame(seed=11)
    p = _args(shipped_cfar, alg)
    want = _oracle(img, alg, p)
    try:
        for variant, tile in [(1, 0), (2, 0), (2, 52), (2, 208), (2, 1024), (3, 0), (3, 156)]:
            ctx._check(ctx.lib.sfe_cfar_set_tuning(ctx.handle, tile, variant))
            assert np.array_equal(FN[alg](img, *p), want), (variant, tile)
    finally:
        ctx._check(ctx.lib.sfe_cfar_set_tuning(ctx.handle, 0, 0))


def test_structured_frames_and_extremes(shipped_cfar):
    p = _args(shipped_cfar, "SOCA")
    frames = [synth.sonar_frame(seed=s) for s in range(3)]
    frames += [np.zeros((1024, 512), np.uint8), np.full((1024, 512), 255, np.uint8)]
    step = np.full((1024, 512), 12, np.uint8)
    step[500:] = 240                      # step exactly across guard/train boundaries
    step[300:306] = 255
    frames.append(step)
    for img in frames:
        assert np.array_equal(cfar.soca(img, *p), _oracle(img, "SOCA", p))


@pytest.mark.parametrize("alg", ALGS)
def test_threshold_maps(alg, shipped_cfar):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (200, 64), dtype=np.uint8)
    p = _args(shipped_cfar, alg)
    m, t = FN2[alg](img, *p)
    mo, to = _oracle(img, alg, p, thr=True)
    assert np.array_equal(m, mo)
    assert t.dtype == np.float32 and np.array_equal(t, to)


@pytest.mark.parametrize("alg", ALGS)
def test_other_windows_and_taus(alg):
    rng = np.random.default_rng(17)
    img = rng.integers(0, 256, (150, 96), dtype=np.uint8)
    for th, gh, tau, k in [(3, 0, 1.3, 2), (8, 2, 0.9, 5), (20, 5, 2.7490637, 39), (30, 1, 4.0, 0)]:
        p = (th, gh, k, tau) if alg == "OS" else (th, gh, tau)
        assert np.array_equal(FN[alg](img, *p), oracle.cfar(img, alg, th, gh, tau, k=k)), (th, gh)


@pytest.mark.parametrize("alg", ALGS)
def test_float_images_take_the_float_path(alg):
    rng = np.random.default_rng(23)
    img = rng.gamma(2.0, 11.3, (140, 50)).astype(np.float32)
    p = (6, 2, 4, 1.7) if alg == "OS" else (6, 2, 1.7)
    m, t = FN2[alg](img, *p)
    mo, to = oracle.cfar(img, alg, 6, 2, 1.7, k=4, want_threshold=True)
    assert np.array_equal(m, mo) and np.array_equal(t, to)
    # float64 input is cast to float32 like the pybind Eigen caster does
    assert np.array_equal(FN[alg](img.astype(np.float64), *p), mo)


@pytest.mark.parametrize("shape", [(1024, 512), (300, 260), (131, 128), (64, 516), (52, 4)])
def test_os_behind_a_gate_counts_candidates_only(shape, monkeypatch):
    """OS-CFAR with the intensity gate of feature_extraction.py:224 goes through cfar_u8_os_gated (only the pixels above
    the gate are looked at: at least k + 1 training cells <= L[x]); same masks as the oracle and as the histogram kernel,
    for shipped and odd windows, gates from 0 (nearly every pixel a candidate) to 250, tiles that end inside the image,
    all-255 and all-0 frames, k at both ends"""
    rows, cols = shape
    monkeypatch.setenv("SFE_CFAR_OS_GATED_MIN", "0")      # every gate through the candidates-only kernel
    rng = np.random.default_rng(rows * 7 + cols)
    first = (synth.sonar_frame(seed=rows + cols, rows=rows, cols=cols, n_blobs=12) if rows > 61 and cols > 11
             else rng.integers(40, 256, (rows, cols)).astype(np.uint8))
    imgs = [first, rng.integers(0, 256, (rows, cols)).astype(np.uint8), np.full((rows, cols), 255, np.uint8),
            np.zeros((rows, cols), np.uint8)]
    for (th, gh, k, tau) in ((20, 5, 10, 9.137608674642355), (6, 2, 0, 1.3), (6, 2, 11, 0.9), (3, 0, 5, 2.0)):
        for gate in (0, 20, 65, 250):   # (0 and 20 take the histogram kernel by default, the others the candidates-only one)
            for img in imgs:
                want = oracle.gate(img, oracle.cfar(img, "OS", th, gh, tau, k), gate)
                got = cfar.detect_gated(img, "OS", (th, gh, k, tau), gate)
                assert np.array_equal(got, want), (shape, th, gh, k, gate)
    monkeypatch.setenv("SFE_CFAR_NO_OS_GATED", "1")     # A/B: the histogram kernel gives the same
    img = imgs[0]
    assert np.array_equal(cfar.detect_gated(img, "OS", (20, 5, 10, 9.137608674642355), 65),
                          oracle.gate(img, oracle.cfar(img, "OS", 20, 5, 9.137608674642355, 10), 65))


@pytest.mark.parametrize("shape", [(1024, 512), (300, 260), (131, 128), (64, 516), (52, 4)])
def test_os_without_a_gate_is_pre_filtered_by_a_window_count(shape, monkeypatch):
    """cfar.os() with NO gate (the drop-in's plain call) and with low gates goes through the pre-filtered candidate kernel
    (round 6: a pixel is looked at when x >= x_hi, or when it can fire at all and its window holds k + 1 cells <= l0).  Same
    masks as the oracle for sonar frames, uniform noise (every pixel a candidate of one kind or the other), dark frames with
    bright pixels (every window passes the count), all-255 / all-0 frames, levels l0 from 0 up (SFE_CFAR_OS_PREF_X), k at
    both ends, odd windows -- and the same as the sliding-histogram kernel (SFE_CFAR_NO_OS_PREF)."""
    rows, cols = shape
    rng = np.random.default_rng(rows * 11 + cols)
    first = (synth.sonar_frame(seed=rows + cols + 1, rows=rows, cols=cols, n_blobs=12) if rows > 61 and cols > 11
             else rng.integers(40, 256, (rows, cols)).astype(np.uint8))
    dark = rng.integers(0, 6, (rows, cols)).astype(np.uint8)          # windows full of cells <= l0 ...
    dark[rng.random((rows, cols)) < 0.03] = 200                       # ... around pixels that can fire
    steps = np.repeat(np.arange(rows, dtype=np.uint8)[:, None], cols, axis=1)   # a ramp: the count changes every row
    imgs = [first, rng.integers(0, 256, (rows, cols)).astype(np.uint8), dark, steps, np.full((rows, cols), 255, np.uint8),
            np.zeros((rows, cols), np.uint8)]
    for (th, gh, k, tau) in ((20, 5, 10, 9.137608674642355), (6, 2, 0, 1.3), (6, 2, 11, 0.9), (3, 0, 5, 2.0), (20, 5, 39, 40.0)):
        for px in (None, "1", "255"):                                  # the level's pixel value: default 80, lowest, highest
            if px is None:
                monkeypatch.delenv("SFE_CFAR_OS_PREF_X", raising=False)
            else:
                monkeypatch.setenv("SFE_CFAR_OS_PREF_X", px)
            for gate in (None, 0, 20):
                for img in imgs:
                    want = oracle.cfar(img, "OS", th, gh, tau, k)
                    if gate is None:
                        got = cfar.os(img, th, gh, k, tau)
                    else:
                        want = oracle.gate(img, want, gate)
                        got = cfar.detect_gated(img, "OS", (th, gh, k, tau), gate)
                    assert np.array_equal(got, want), (shape, th, gh, k, tau, px, gate)
    monkeypatch.delenv("SFE_CFAR_OS_PREF_X", raising=False)
    monkeypatch.setenv("SFE_CFAR_NO_OS_PREF", "1")      # A/B: the histogram kernel gives the same
    assert np.array_equal(cfar.os(imgs[0], 20, 5, 10, 9.137608674642355), oracle.cfar(imgs[0], "OS", 20, 5, 9.137608674642355, 10))


def test_fused_intensity_gate(shipped_cfar):
    img = synth.sonar_frame(seed=2)
    for alg in ALGS:
        p = _args(shipped_cfar, alg)
        want = oracle.gate(img, _oracle(img, alg, p), 65)
        assert np.array_equal(cfar.detect_gated(img, alg, p, 65), want)
    assert want.sum() > 0


def test_os_requires_integer_rank():
    with pytest.raises(TypeError):
        cfar.os(np.zeros((60, 4), np.uint8), 20, 5, 20.0, 1.0)


def test_batched_device_path_full_size(shipped_cfar, ctx):
    """BASELINE config sizes, device-resident batch: every frame equals the oracle, and the
    batch is translation invariant (frame f of a batch == the same frame processed alone)."""
    n, rows, cols = 6, 1024, 512
    frames = np.stack([synth.sonar_frame(seed=100 + s) for s in range(n)])
    th, gh, tau = shipped_cfar.params["SOCA"]
    d_img, d_mask = ctx.alloc(frames.nbytes), ctx.alloc(frames.nbytes)
    d_img.upload(frames)
    ctx._check(ctx.lib.sfe_cfar_u8_batch_dev(ctx.handle, d_img.ptr, n, rows, cols, 1, th, gh, 0, tau, 65,
                                             d_mask.ptr, None))
    ctx.sync()
    got = d_mask.download(np.uint8, frames.size).reshape(frames.shape)
    for f in range(n):
        want = oracle.gate(frames[f], oracle.cfar(frames[f], "SOCA", th, gh, tau), 65)
        assert np.array_equal(got[f], want), f
    # high-res Oculus config (2048 x 1024)
    big = synth.sonar_frame(seed=7, rows=2048, cols=1024, n_blobs=120)
    assert np.array_equal(cfar.soca(big, th, gh, tau), oracle.cfar(big, "SOCA", th, gh, tau))
    d_img.free()
    d_mask.free()


@pytest.mark.parametrize("alg", ["CA", "SOCA", "GOCA"])
@pytest.mark.parametrize("shape,frames", [((2048, 1024), 3), ((1024, 512), 19), ((157, 256), 9), ((104, 260), 8)])
def test_ring_kernel_xcd_map_and_march_direction(alg, shape, frames, shipped_cfar, ctx):
    """The ring kernel maps frame f to XCD f % 8 (frame count padded to 8) and marches even tiles up,
    odd tiles down: batches whose frame count is not a multiple of 8, the config-B frame shape, tile
    counts that are odd / a single shifted tile -- every frame must equal the oracle."""
    rng = np.random.default_rng(frames * 1000 + shape[0])
    imgs = np.stack([synth.sonar_frame(seed=int(s), rows=shape[0], cols=shape[1], n_blobs=20)
                     for s in rng.integers(0, 1 << 30, frames)])
    p = _args(shipped_cfar, alg)
    d_in, d_out = ctx.alloc(imgs.nbytes), ctx.alloc(imgs.nbytes)
    d_in.upload(imgs)
    from sonar_slam_amd import _lib
    ctx._check(ctx.lib.sfe_cfar_u8_batch_dev(ctx.handle, d_in.ptr, frames, shape[0], shape[1], _lib.ALG[alg], p[0], p[1],
                                             0, float(p[-1]), -1, d_out.ptr, None))
    got = d_out.download(np.uint8, imgs.size).reshape(imgs.shape)
    for f in range(frames):
        assert np.array_equal(got[f], _oracle(imgs[f], alg, p)), f
    d_in.free()
    d_out.free()


def test_hip_cfar_equals_the_reference_fixture():
    """every HIP CFAR path (ring, generic, float image, threshold maps) against masks and float threshold maps the
    reference's own cfar.cpp produced (tests/golden/cfar_ref.npz); needs neither /root/reference nor oracle/_ref"""
    from test_golden import _cfar_ref_cases
    from sonar_slam_amd import cfar
    n = 0
    for key, img, alg, th, gh, tau, k, mask, thr in _cfar_ref_cases():
        fn, fn2 = getattr(cfar, alg.lower()), getattr(cfar, alg.lower() + "2")
        got = fn(img, th, gh, k, tau) if alg == "OS" else fn(img, th, gh, tau)
        got2 = fn2(img, th, gh, k, tau) if alg == "OS" else fn2(img, th, gh, tau)
        assert np.array_equal(got, mask), key
        assert np.array_equal(got2[0], mask) and np.array_equal(got2[1], thr), key
        n += 1
    assert n == 48


def _bits_call(ctx, frames, alg, p, gate):
    """sfe_cfar_u8_bits_batch_dev on a stack of frames -> (0/1 masks unpacked on the host, pad words)"""
    n, rows, cols = frames.shape
    wpf = (rows * cols + 31) // 32 + 1
    d_img, d_bits = ctx.alloc(frames.nbytes), ctx.alloc(n * wpf * 4)
    try:
        d_img.upload(frames)
        d_bits.upload(np.full(n * wpf, 0xDEADBEEF, np.uint32))     # every word must be written by the call
        k = p[2] if alg == "OS" else 0
        ctx._check(ctx.lib.sfe_cfar_u8_bits_batch_dev(ctx.handle, d_img.ptr, n, rows, cols, _lib.ALG[alg], p[0], p[1],
                                                      k, float(p[-1]), gate, d_bits.ptr))
        ctx.sync()
        w = d_bits.download(np.uint32, n * wpf).reshape(n, wpf)
    finally:
        d_img.free()
        d_bits.free()
    px = rows * cols
    masks = np.stack([np.unpackbits(r.view(np.uint8), bitorder="little")[:px].reshape(rows, cols) for r in w])
    tail = np.stack([np.unpackbits(r.view(np.uint8), bitorder="little")[px:] for r in w])
    return masks, tail


@pytest.mark.parametrize("alg", ALGS)
@pytest.mark.parametrize("window", [(20, 5), (16, 4), (10, 2), (8, 1), (12, 3)])
@pytest.mark.parametrize("shape", [(1024, 512), (200, 256), (130, 288), (96, 64), (70, 100)])
def test_bit_stream_output_equals_the_oracle_mask(ctx, alg, window, shape):
    """The BITS ring kernel ((20,5) .. (8,1) on whole-word rows) and the pack fallback (every other call) write the
    same detections as the byte kernels: bit iy*cols+ix of the frame's stream, pad bits 0."""
    rng = np.random.default_rng(abs(hash((alg, window, shape))) % 2**32)
    frames = rng.integers(0, 256, (3,) + shape, dtype=np.uint8)
    frames[1] = synth.sonar_frame(seed=5)[:shape[0], :shape[1]] if shape[1] <= 512 else frames[1]
    p = (window[0], window[1], window[0] + 3, 1.2) if alg == "OS" else (window[0], window[1], 1.1)
    for gate in (-1, 65):
        masks, tail = _bits_call(ctx, frames, alg, p, gate)
        assert not tail.any()
        for f in range(len(frames)):
            want = _oracle(frames[f], alg, p)
            if gate >= 0:
                want = oracle.gate(frames[f], want, gate)
            assert np.array_equal(masks[f], want), (gate, f)


def test_bit_stream_output_many_frames_shipped_window(ctx, shipped_cfar):
    """more frames than XCDs, the frame count not a multiple of 8, structured frames and the extremes"""
    p = _args(shipped_cfar, "SOCA")
    frames = [synth.sonar_frame(seed=40 + s) for s in range(17)]
    frames += [np.zeros((1024, 512), np.uint8), np.full((1024, 512), 255, np.uint8)]
    frames = np.stack(frames)
    masks, tail = _bits_call(ctx, frames, "SOCA", p, 65)
    assert not tail.any()
    for f in range(len(frames)):
        assert np.array_equal(masks[f], oracle.gate(frames[f], _oracle(frames[f], "SOCA", p), 65)), f
