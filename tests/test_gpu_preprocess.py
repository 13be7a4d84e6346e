"""Pre-process kernels through the reference-named C API (cuda_preprocess.h:50-112) vs the scalar oracle."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(pkg):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from unina_yolo_dla_amd import engine
    return torch, engine.load_library(), engine


def test_norm_params_and_allocators(env):
    torch, L, engine = env
    p = L.create_norm_params_imagenet()
    assert (p.mean_r, p.std_b) == (pytest.approx(0.485), pytest.approx(0.225))          # cuda_preprocess.cu:73-75
    q = L.create_norm_params(0.0, 0.0, 0.0, 1.0, 1.0, 1.0)
    assert (q.mean_g, q.std_r) == (0.0, 1.0)
    buf = L.allocate_preprocess_buffer(640, 640)
    assert buf
    L.free_preprocess_buffer(buf)
    assert not L.allocate_preprocess_buffer(0, 10)                                        # NULL on failure
    s = L.create_preprocess_stream()
    assert s
    L.destroy_preprocess_stream(s)


@pytest.mark.parametrize("src,dst", [((720, 1280), (640, 640)), ((480, 640), (640, 640)), ((64, 96), (64, 96)),
                                     ((50, 70), (33, 45))])   # (row tails: widths that are not multiples of the 4-pixel quads)
def test_bgra_resize_and_plain_match_oracle(env, oracle_mod, src, dst):
    torch, L, engine = env
    rng = np.random.default_rng(11)
    sh, sw = src
    pitch = ((sw * 4 + 255) // 256) * 256                                                 # node requires pitch % 256 == 0 (:590-596)
    host = rng.integers(0, 256, (sh, pitch), dtype=np.uint8)
    d_in = torch.from_numpy(host).cuda()
    norm = L.create_norm_params_imagenet()
    stream = torch.cuda.current_stream().cuda_stream
    out = torch.empty((3, dst[0], dst[1]), dtype=torch.float32, device="cuda")
    assert L.preprocess_bgra_resize(d_in.data_ptr(), out.data_ptr(), sw, sh, pitch, dst[1], dst[0], norm, stream) == 0
    torch.cuda.synchronize()
    img = np.lib.stride_tricks.as_strided(host, shape=(sh, sw, 4), strides=(pitch, 4, 1))
    want = oracle_mod.preprocess_bgra(np.ascontiguousarray(img), dst_hw=dst)
    np.testing.assert_array_equal(out.cpu().numpy(), want)                                # same expression tree, no FMA contraction
    out2 = torch.empty((3, sh, sw), dtype=torch.float32, device="cuda")
    assert L.preprocess_bgra(d_in.data_ptr(), out2.data_ptr(), sw, sh, pitch, norm, stream) == 0
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out2.cpu().numpy(), oracle_mod.preprocess_bgra(np.ascontiguousarray(img)))
    assert L.preprocess_bgra(d_in.data_ptr(), out2.data_ptr(), sw, sh, sw * 4 - 4, norm, stream) != 0   # pitch too small


@pytest.mark.parametrize("h,w,pitch", [(360, 640, 768), (50, 70, 74)])   # (the second: row tails and an unaligned pitch)
def test_nv12_matches_oracle(env, oracle_mod, h, w, pitch):
    torch, L, engine = env
    rng = np.random.default_rng(12)
    y = rng.integers(0, 256, (h, pitch), dtype=np.uint8)
    uv = rng.integers(0, 256, (h // 2, pitch), dtype=np.uint8)
    dy, duv = torch.from_numpy(y).cuda(), torch.from_numpy(uv).cuda()
    out = torch.empty((3, h, w), dtype=torch.float32, device="cuda")
    assert L.preprocess_nv12(dy.data_ptr(), duv.data_ptr(), out.data_ptr(), w, h, pitch, pitch,
                             L.create_norm_params_imagenet(), torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    want = oracle_mod.preprocess_nv12(np.ascontiguousarray(y[:, :w]), np.ascontiguousarray(uv[:, :w]))
    np.testing.assert_array_equal(out.cpu().numpy(), want)


def test_camera_to_detections_pipeline(env, pkg, sd7):
    """processGpuBuffer end to end (perception_node.cpp:601-656): BGRA camera buffer -> resize/normalise -> engine ->
    detections, all on the GPU."""
    torch, L, engine = env
    e = engine.Engine.from_state_dict(sd7)
    try:
        rng = np.random.default_rng(13)
        cam = torch.from_numpy(rng.integers(0, 256, (720, 1280 * 4), dtype=np.uint8)).cuda()
        images = torch.empty((1, 3, 640, 640), dtype=torch.float32, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        assert L.preprocess_bgra_resize(cam.data_ptr(), images.data_ptr(), 1280, 720, 1280 * 4, 640, 640,
                                        L.create_norm_params_imagenet(), s) == 0
        dets = e.infer(images, 0.5, 0.45, 0.1)
        assert dets.dtype == engine.DET_DTYPE and np.all(np.diff(dets["confidence"]) <= 0)
    finally:
        e.close()


@pytest.mark.parametrize("cam_hw", [(720, 1280), (640, 640), (480, 600)])
def test_camera_frame_in_the_stem_kernel_is_bit_identical(env, pkg, sd7, cam_hw):
    """unina_infer_bgra: the pre-process computed inside the stem kernel (no fp32 tensor between camera frame and
    network) against the two-step form preprocess_bgra[_resize] + unina_infer: same arithmetic, so the stem output and
    the detections must agree bit for bit. 720p (down-scale), the network's own size (no resize) and an up-scaled,
    pitched frame."""
    torch, L, engine = env
    ch, cw = cam_hw
    pitch = cw * 4 + 64
    e = engine.Engine.from_state_dict(sd7)
    try:
        rng = np.random.default_rng(17)
        host = rng.integers(0, 256, (ch, pitch), dtype=np.uint8)
        # a smooth pattern under the noise so that some cells pass the threshold either way
        cam = torch.from_numpy(host).cuda()
        images = torch.empty((1, 3, 640, 640), dtype=torch.float32, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        norm = L.create_norm_params_imagenet()
        if (ch, cw) == (640, 640):
            assert L.preprocess_bgra(cam.data_ptr(), images.data_ptr(), cw, ch, pitch, norm, s) == 0
        else:
            assert L.preprocess_bgra_resize(cam.data_ptr(), images.data_ptr(), cw, ch, pitch, 640, 640, norm, s) == 0
        want = e.infer(images, 0.3, 0.45, 0.1)
        e.set_fusion(False)
        e.forward(images)
        stem_want = e.read_buffer("backbone.stem")
        e.set_fusion(True)
        got = e.infer_bgra(cam, cw, ch, pitch, norm, 0.3, 0.45, 0.1)
        assert len(want) > 0 and got.tobytes() == want.tobytes()
        # the tensor path still works afterwards (the stem node is re-pointed back), and the camera path again
        assert e.infer(images, 0.3, 0.45, 0.1).tobytes() == want.tobytes()
        assert e.infer_bgra(cam, cw, ch, pitch, norm, 0.3, 0.45, 0.1).tobytes() == want.tobytes()
        assert np.array_equal(e.read_buffer("backbone.stem"), stem_want)
    finally:
        e.close()
