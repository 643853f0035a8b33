"""GPU parity of the decode / render path against the CPU oracle (reference-shaped loop over the C restatement of
the reference's kernels + PyTorch-CPU grid_sample / Linear decode).

Tolerances (fp32 path): sigma relative 2e-5; rgb absolute 2e-6 per sample; rendered RGB absolute 2e-5
(PSNR-equivalent > 90 dB, far inside the north-star's 1e-4); depth absolute 1e-4.  Integer contract: per-ray
sample counts equal to the oracle's, except rays whose termination test T < 1e-4 sits within float noise of
the threshold (the reference itself uses the hardware __expf there): at most 1 ray in 2000 may differ, by
construction never by more than the samples of one reference iteration."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene():
    from oracle import render as R
    from ssdnerf_amd import synthetic as S
    params = S.make_decoder_params()
    code = S.make_triplane()
    g = torch.Generator().manual_seed(7)
    jit = [torch.rand(64 ** 3, 3, generator=g).numpy() for _ in range(8)]
    grid, bits, th = R.get_density(params, code, jit, density_thresh=0.1)
    return dict(params=params, code=code, jitters=jit, grid=grid, bits=bits, thresh=th)


@pytest.fixture(scope="module")
def decoder(scene):
    from ssdnerf_amd.decoders import TriPlaneDecoder
    dec = TriPlaneDecoder(base_layers=[18, 64], density_layers=[64, 1], color_layers=[64, 3], dir_layers=[16, 64], max_steps=256)
    missing = dec.load_state_dict(scene["params"], strict=False)
    assert set(missing.missing_keys) <= {"aabb"} and not missing.unexpected_keys      # reference state-dict keys load as-is
    return dec.cuda().eval()


def _view(idx, size=128):
    from oracle import render as R
    from ssdnerf_amd import synthetic as S
    ro, rd = R.get_cam_rays(S.spiral_poses()[idx][None], S.cars_intrinsics(size, size)[None], size, size)
    return ro.reshape(-1, 3).numpy(), rd.reshape(-1, 3).numpy()


def test_point_decode_matches_oracle(scene, decoder):
    from oracle.decoder import point_decode
    g = torch.Generator().manual_seed(3)
    xyz = (torch.rand(50000, 3, generator=g) * 2 - 1)
    xyz[:100] = torch.tensor([1.0, -1.0, 1.0])              # border texels
    xyz[100:200] *= 0.0
    dirs = torch.nn.functional.normalize(torch.randn(50000, 3, generator=g), dim=-1)
    sig0, rgb0 = point_decode(scene["params"], scene["code"], xyz, dirs)
    code = scene["code"].cuda()[None]
    with torch.no_grad():
        sig, rgb, n = decoder.point_decode([xyz.cuda()], [dirs.cuda()], code)
        sig_e, rgb_e, _ = decoder.point_decode_eager([xyz.cuda()], [dirs.cuda()], code)
        sd, nd = decoder.point_density_decode([xyz.cuda()], code)
    assert n == [50000] and nd == [50000]
    np.testing.assert_allclose(sig.cpu().numpy(), sig0.numpy(), rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(sd.cpu().numpy(), sig0.numpy(), rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(rgb.cpu().numpy(), rgb0.numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(sig_e.cpu().numpy(), sig0.numpy(), rtol=5e-5, atol=1e-7)     # eager ROCm vs CPU (baseline B1 sanity)
    np.testing.assert_allclose(rgb_e.cpu().numpy(), rgb0.numpy(), rtol=0, atol=5e-6)
    # empty and ragged inputs
    e = torch.zeros(0, 3).cuda()
    with torch.no_grad():
        s2, c2, n2 = decoder.point_decode([e, xyz[:77].cuda()], [e, dirs[:77].cuda()], code.expand(2, -1, -1, -1, -1).contiguous())
        s3, c3, n3 = decoder.point_decode_eager([e, xyz[:77].cuda()], [e, dirs[:77].cuda()], code.expand(2, -1, -1, -1, -1).contiguous())
    assert n3 == [0, 77] and torch.allclose(s3, s2, rtol=1e-4, atol=1e-7)
    assert n2 == [0, 77] and s2.shape == (77,) and c2.shape == (77, 3)


def test_point_decode_fp16_planes(scene, decoder):
    from oracle.decoder import point_decode
    from ssdnerf_amd.decoders import pack_triplanes
    from ssdnerf_amd import _cabi as C
    g = torch.Generator().manual_seed(4)
    xyz = (torch.rand(20000, 3, generator=g) * 2 - 1)
    dirs = torch.nn.functional.normalize(torch.randn(20000, 3, generator=g), dim=-1)
    code16 = scene["code"].half()
    sig0, rgb0 = point_decode(scene["params"], code16.float(), xyz, dirs)                # fp16-rounded code, fp32 math
    planes = pack_triplanes(code16.cuda()[None], torch.float16)
    sig, rgb = torch.empty(20000).cuda(), torch.empty(20000, 3).cuda()
    x, d = xyz.cuda().contiguous(), dirs.cuda().contiguous()
    C.check(C.lib().ssdnerf_point_decode(C.ptr(planes[0]), 1, C.u32(128), C.u32(128), C.ptr(decoder.packed_params()), C.ptr(x), C.ptr(d),
                                         C.u32(20000), C.f32(0.001), C.ptr(sig), C.ptr(rgb), C.stream()), "point_decode")
    np.testing.assert_allclose(sig.cpu().numpy(), sig0.numpy(), rtol=3e-5, atol=1e-7)
    np.testing.assert_allclose(rgb.cpu().numpy(), rgb0.numpy(), rtol=0, atol=3e-6)


def _render_gpu(decoder, scene, ro, rd, mode, dt_gamma=0.0, counts=False):
    from ssdnerf_amd.decoders import pack_triplanes
    code = scene["code"].cuda()[None]
    bits = torch.from_numpy(scene["bits"]).cuda()[None]
    o, d = torch.from_numpy(ro).cuda()[None], torch.from_numpy(rd).cuda()[None]
    with torch.no_grad():
        if mode == "fused":
            planes = pack_triplanes(code)
            out = decoder.render_packed(planes, o, d, bits, [64], [dt_gamma], 1e-4, bg_color=1.0, want_counts=True)
            rgb = out["image"][0].cpu().numpy()
            cnt = decoder.last_render_stats["sample_counts"][0].cpu().numpy()
            return rgb, out["depth"][0].cpu().numpy(), out["weights_sum"][0].cpu().numpy(), cnt
        decoder.render_mode = "stepwise"
        out = decoder(o, d, code, bits, 64, dt_gamma=dt_gamma, perturb=False)
        decoder.render_mode = "fused"
        ws = out["weights_sum"][0]
        rgb = (out["image"][0] + 1.0 * (1 - ws.unsqueeze(-1))).cpu().numpy()
        return rgb, out["depth"][0].cpu().numpy(), ws.cpu().numpy(), decoder.last_render_stats["iterations"][0]


@pytest.mark.parametrize("view,dt_gamma", [(64, 0.0), (200, 0.0038095), (5, 0.0)])
def test_render_view_fused_and_stepwise_vs_oracle(scene, decoder, view, dt_gamma):
    from oracle import render as R
    ro, rd = _view(view)
    tr = {}
    rgb0, dep0, ws0 = R.render_eval(scene["params"], scene["code"], scene["bits"], ro, rd, dt_gamma=dt_gamma, trace=tr)
    assert tr["samples_marched"].sum() > 10000
    # reference-shaped loop over the unfused HIP ops: same iteration history (n_alive, n_step), modulo threshold-noise rays
    rgb1, dep1, ws1, hist = _render_gpu(decoder, scene, ro, rd, "stepwise", dt_gamma)
    assert len(hist) == len(tr["iterations"])
    assert all(abs(a[0] - b[0]) <= 8 and a[1] == b[1] for a, b in zip(hist, tr["iterations"])), (hist, tr["iterations"])
    np.testing.assert_allclose(rgb1, rgb0, rtol=0, atol=2e-5)
    np.testing.assert_allclose(dep1, dep0, rtol=0, atol=1e-4)
    # fused kernel
    rgb2, dep2, ws2, cnt = _render_gpu(decoder, scene, ro, rd, "fused", dt_gamma)
    np.testing.assert_allclose(rgb2, rgb0, rtol=0, atol=2e-5)
    np.testing.assert_allclose(dep2, dep0, rtol=0, atol=1e-4)
    np.testing.assert_allclose(ws2, ws0, rtol=0, atol=1e-5)
    mse = float(((rgb2 - rgb0) ** 2).mean())
    assert mse < 1e-10
    # the two VALU pipelines (two-stage hit queue vs single persistent kernel) run the same arithmetic: bit-identical;
    # the default MFMA shading kernel sums the same products in another fixed order: fp32-rounding close, same counts
    # except on rays sitting at the T_thresh boundary
    res = {}
    for pipe in ("single", "queue"):
        decoder.fused_pipeline = pipe
        try:
            res[pipe] = _render_gpu(decoder, scene, ro, rd, "fused", dt_gamma)
        finally:
            decoder.fused_pipeline = "queue_mfma"
    assert np.array_equal(res["single"][3], res["queue"][3])
    assert np.array_equal(res["single"][0].view(np.uint32), res["queue"][0].view(np.uint32))
    assert np.array_equal(res["single"][1].view(np.uint32), res["queue"][1].view(np.uint32))
    np.testing.assert_allclose(res["queue"][0], rgb0, rtol=0, atol=2e-5)
    np.testing.assert_allclose(rgb2, res["queue"][0], rtol=0, atol=1e-5)
    assert int((cnt != res["queue"][3]).sum()) <= max(1, cnt.size // 2000)


def test_fused_sample_counts_match_oracle_composited(scene, decoder):
    """Per-ray number of samples that enter the composite: bit-exact vs a per-ray serial statement over the oracle ops."""
    import oracle
    from oracle.decoder import point_decode
    o = oracle.ops()
    ro, rd = _view(64)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = o.near_far_from_aabb(ro, rd, aabb, 0.2)
    N = ro.shape[0]
    # march every ray to the end once (n_step = max_steps covers any ray), decode, then composite serially per ray
    alive = np.arange(N, dtype=np.int32)
    xyzs, dirs, deltas = o.march_rays(N, 256, alive, nears.copy(), ro, rd, 1.0, scene["bits"], 1, 64, nears, fars, dt_gamma=0.0, max_steps=256)
    with torch.no_grad():
        sig, rgb = point_decode(scene["params"], scene["code"], torch.from_numpy(xyzs), torch.from_numpy(dirs))
    sig = sig.numpy().reshape(N, 256); dl = deltas.reshape(N, 256, 2)
    want = np.zeros(N, np.int64)
    near_thresh = np.zeros(N, bool)
    for n in range(N):
        ws = np.float32(0)
        k = 0
        while k < 256 and dl[n, k, 0] != 0:
            alpha = np.float32(1) - np.exp(-sig[n, k] * dl[n, k, 0], dtype=np.float32)
            T = np.float32(1) - ws
            ws = ws + alpha * T
            k += 1
            if abs(float(T) - 1e-4) < 2e-6:
                near_thresh[n] = True
            if T < 1e-4:
                break
        want[n] = k
    _, _, _, cnt = _render_gpu(decoder, scene, ro, rd, "fused")
    diff = cnt != want
    assert (diff & ~near_thresh).sum() == 0, "sample counts differ on rays that are not at the T_thresh boundary"
    assert diff.sum() <= max(1, N // 2000)
    assert want.sum() > 10000 and int(decoder.last_render_stats["overflow"].item()) == 0


def test_fused_render_edge_cases(scene, decoder):
    from ssdnerf_amd.decoders import pack_triplanes
    code = scene["code"].cuda()[None]
    planes = pack_triplanes(code)
    bits = torch.from_numpy(scene["bits"]).cuda()[None]
    # rays that all miss the box, a single ray, and a ragged (non multiple of 64/256) count
    for n in (1, 63, 257, 1000):
        o = torch.tensor([[0.0, 0.0, 3.0]]).repeat(n, 1).cuda()
        d = torch.tensor([[0.0, 1.0, 0.0]]).repeat(n, 1).cuda()
        for rays in (([o], [d]), (o[None], d[None])):            # per-scene lists (single kernel) and dense batch (hit queue)
            out = decoder.render_packed(planes, rays[0], rays[1], bits, [64], [0.0], 1e-4, bg_color=1.0, want_counts=True)
            assert torch.all(out["image"][0] == 1.0) and torch.all(out["weights_sum"][0] == 0) and torch.all(out["depth"][0] == 0)
            assert int(decoder.last_render_stats["sample_counts"][0].abs().sum()) == 0
    # empty bitfield: nothing is ever sampled
    zero_bits = torch.zeros_like(bits)
    from oracle import render as R
    ro, rd = _view(30)
    out = decoder.render_packed(planes, torch.from_numpy(ro).cuda()[None], torch.from_numpy(rd).cuda()[None], zero_bits, [64], [0.0], 1e-4, bg_color=0.5)
    assert torch.all(out["image"][0] == 0.5)
    # full bitfield on the fog scene hits the step cap -> overflow is reported, never silent
    full = torch.full_like(bits, 255)
    decoder.render_packed(planes, torch.from_numpy(ro).cuda()[None], torch.from_numpy(rd).cuda()[None], full, [64], [0.0], 0.0, bg_color=1.0, check_overflow=False)
    assert int(decoder.last_render_stats["overflow"].item()) >= 0


def test_multi_scene_batch_matches_per_scene(decoder):
    """S scenes in one launch (per-scene queues, XCD-pinned shading) == the same scenes rendered one by one."""
    from ssdnerf_amd import synthetic as S
    from ssdnerf_amd.decoders import pack_triplanes
    from ssdnerf_amd.density import get_density
    for ns in (3, 8):
        code = S.make_scene_batch(ns, seed=77).cuda()
        g = torch.Generator().manual_seed(11)
        jit = [torch.rand(64 ** 3, 3, generator=g).cuda() for _ in range(2)]
        _, bits = get_density(decoder, code, 64, density_thresh=0.1, density_step=2, jitters=jit)
        planes = pack_triplanes(code)
        ro, rd = _view(100)
        o = torch.from_numpy(ro).cuda()[None].expand(ns, -1, -1).contiguous()
        d = torch.from_numpy(rd).cuda()[None].expand(ns, -1, -1).contiguous()
        gam = torch.linspace(0.0, 0.004, ns).cuda()
        out = decoder.render_packed(planes, o, d, bits, 64, gam, 1e-4, bg_color=1.0, want_counts=True)
        cn = decoder.last_render_stats["sample_counts"]
        for s in range(ns):
            one = decoder.render_packed(planes[s:s + 1], o[s:s + 1], d[s:s + 1], bits[s:s + 1], 64, [float(gam[s])], 1e-4, bg_color=1.0, want_counts=True)
            assert torch.equal(one["image"][0], out["image"][s]) and torch.equal(one["depth"][0], out["depth"][s])
            assert torch.equal(decoder.last_render_stats["sample_counts"][0], cn[s])
        assert int(cn.sum()) > 1000


def test_density_grid_update_matches_oracle(scene, decoder):
    from ssdnerf_amd.density import update_density_grid
    from oracle import render as R
    code = scene["code"].cuda()[None]
    for dtype, np_dtype in ((torch.float32, np.float32), (torch.float16, np.float16)):
        grid0 = np.zeros(64 ** 3, np_dtype)
        grid1 = torch.zeros(1, 64 ** 3, dtype=dtype).cuda()
        bits1 = torch.zeros(1, 64 ** 3 // 8, dtype=torch.uint8).cuda()
        for it, jit in enumerate(scene["jitters"][:3]):
            decay = 1.0 if it < 2 else 0.9
            b0, th0 = R.update_extra_state(scene["params"], scene["code"], grid0, jit, 64, density_thresh=0.1, decay=decay)
            th1 = update_density_grid(decoder, code, grid1, bits1, density_thresh=0.1, decay=decay, jitter=torch.from_numpy(jit).cuda())
        g1 = grid1[0].float().cpu().numpy()
        np.testing.assert_allclose(g1, grid0.astype(np.float32), rtol=2e-3 if dtype == torch.float16 else 3e-5, atol=1e-6)
        flips = int(np.unpackbits(bits1[0].cpu().numpy() ^ b0).sum())
        assert flips <= 4, flips                                  # cells within float noise of the threshold
        assert abs(float(th1) - th0) <= 1e-6 + 1e-3 * th0


@pytest.mark.parametrize("dt_gamma", [0.0, 0.0038095])
def test_ticket_order_changes_nothing(scene, decoder, dt_gamma, monkeypatch):
    """r05: the shading kernel takes the hit queue's 64-entry slices longest first (k_ticket_order: counting sort of the slices by the largest step
    bound of their rays, queue in arrival order) instead of front to back over a long / short split queue.  Which wave shades a ray, and when, never
    enters the ray's arithmetic: image, depth, weights and per-ray sample counts of a multi-view camera-fed render are bit-identical with the order
    switched off (SSDNERF_TICKET_ORDER=0) -- and every hitting ray is shaded exactly once (no slice lost or taken twice: equal sample totals)."""
    from ssdnerf_amd import synthetic as S
    from ssdnerf_amd.decoders import pack_triplanes
    planes = pack_triplanes(scene["code"].cuda()[None])
    bits = torch.from_numpy(scene["bits"]).cuda()[None]
    nv = 9
    cams = (S.spiral_poses()[::28][:nv][None].cuda().contiguous(), S.cars_intrinsics(128, 128)[None, None].expand(1, nv, -1).cuda().contiguous(), 128, 128)

    def render():
        with torch.no_grad():
            out = decoder.render_packed(planes, None, None, bits, [64], [dt_gamma], 1e-4, bg_color=1.0, want_counts=True, cams=cams)
        return [out["image"][0].cpu().numpy(), out["depth"][0].cpu().numpy(), out["weights_sum"][0].cpu().numpy(),
                decoder.last_render_stats["sample_counts"][0].cpu().numpy()]
    ordered = render()
    monkeypatch.setenv("SSDNERF_TICKET_ORDER", "0")
    plain = render()
    monkeypatch.delenv("SSDNERF_TICKET_ORDER")
    assert (ordered[3] > 0).sum() > 5000 and int(ordered[3].sum()) == int(plain[3].sum())
    for a, b in zip(ordered, plain):
        assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b)


@pytest.mark.parametrize("n_rays", [1, 63, 64, 65, 5001])
def test_ticket_order_on_ragged_ray_counts(scene, decoder, n_rays, monkeypatch):
    """The ticket order's bookkeeping on ray counts that are not multiples of its 64-entry slices (key stride rounded up, a partial last slice, fewer hits than one
    slice, a single ray): ray-array renders of the first ``n_rays`` rays of a view that looks at the object, bit-identical with the order switched off."""
    ro, rd = _view(64)
    centre = 128 * 64 + 40                                                     # start inside the object's silhouette so that small counts hit something
    o = torch.from_numpy(ro[centre:centre + n_rays] if n_rays < 5000 else ro[:n_rays]).cuda()[None].contiguous()
    d = torch.from_numpy(rd[centre:centre + n_rays] if n_rays < 5000 else rd[:n_rays]).cuda()[None].contiguous()
    from ssdnerf_amd.decoders import pack_triplanes
    planes = pack_triplanes(scene["code"].cuda()[None])
    bits = torch.from_numpy(scene["bits"]).cuda()[None]

    def render():
        with torch.no_grad():
            out = decoder.render_packed(planes, o, d, bits, [64], [0.0], 1e-4, bg_color=1.0, want_counts=True)
        return [out["image"][0].cpu().numpy(), out["depth"][0].cpu().numpy(), decoder.last_render_stats["sample_counts"][0].cpu().numpy()]
    a = render()
    monkeypatch.setenv("SSDNERF_TICKET_ORDER", "0")
    b = render()
    monkeypatch.delenv("SSDNERF_TICKET_ORDER")
    assert a[2].shape == (n_rays,) and (n_rays < 64 or int(a[2].sum()) > 0)
    for x, y in zip(a, b):
        assert np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x, y.view(np.uint32) if y.dtype == np.float32 else y)


@pytest.mark.parametrize("view,dt_gamma", [(64, 0.0), (17, 0.0038095), (230, 0.0)])
def test_coarse_empty_space_pretest_changes_nothing(scene, decoder, view, dt_gamma, monkeypatch):
    """k_ray_cull's conservative coarse-occupancy pre-test (and the view-level tile masks) only removes marches that cannot find a sample: every output, including the
    per-ray sample counts, is bit-identical with the pre-test switched off (SSDNERF_NO_COARSE=1)."""
    ro, rd = _view(view)
    with_pretest = _render_gpu(decoder, scene, ro, rd, "fused", dt_gamma)
    monkeypatch.setenv("SSDNERF_NO_COARSE", "1")
    without = _render_gpu(decoder, scene, ro, rd, "fused", dt_gamma)
    monkeypatch.delenv("SSDNERF_NO_COARSE")
    assert (without[3] == 0).sum() > 1000 and (without[3] > 0).sum() > 1000          # both kinds of rays are present
    for a, b in zip(with_pretest, without):
        assert np.array_equal(np.asarray(a).view(np.uint32) if np.asarray(a).dtype == np.float32 else np.asarray(a), np.asarray(b).view(np.uint32) if np.asarray(b).dtype == np.float32 else np.asarray(b))


def test_cam_rays_kernel_matches_tensor_op_form():
    """ssdnerf_cam_rays (one HIP pass) against the reference-shaped tensor ops on the CPU, and against the golden fixture the reference's own
    get_cam_rays produced (tests/golden/cam_rays_64.npz)."""
    import os
    from ssdnerf_amd import nerf, synthetic as S
    poses = S.spiral_poses(7)[None].expand(2, -1, -1, -1).contiguous()
    intr = S.cars_intrinsics(48, 40)[None, None].expand(2, 7, -1).contiguous()
    o_cpu, d_cpu = nerf.get_cam_rays(poses, intr, 40, 48)
    o_gpu, d_gpu = nerf.get_cam_rays(poses.cuda(), intr.cuda(), 40, 48)
    assert o_gpu.shape == (2, 7, 40, 48, 3)
    assert torch.equal(o_gpu.cpu(), o_cpu.contiguous())
    assert (d_gpu.cpu() - d_cpu).abs().max().item() <= 3e-7                          # a few ulp: summation order / FMA of the 3x3 rotation
    f = np.load(os.path.join(os.path.dirname(__file__), "golden", "cam_rays_64.npz"))
    ro, rd = nerf.get_cam_rays(torch.from_numpy(f["pose"]).cuda(), torch.from_numpy(f["intrinsics"]).cuda(), 64, 64)     # 1 scene, 1 view, 64x64
    assert np.abs(rd.cpu().numpy().reshape(f["rays_d"].shape) - f["rays_d"]).max() <= 3e-7
    assert np.array_equal(ro.cpu().numpy().reshape(f["rays_o"].shape), f["rays_o"])


def test_quantize_u8_kernel_matches_torch():
    from ssdnerf_amd import nerf
    g = torch.Generator().manual_seed(3)
    x = torch.rand(1003, 3, generator=g) * 1.4 - 0.2
    x[:8, 0] = torch.tensor([0.5 / 255, 1.5 / 255, 2.5 / 255, 0.0, 1.0, -1.0, 2.0, 127.5 / 255])       # ties round to even, clamps
    got = nerf.quantize_u8(x.cuda()).cpu()
    want = torch.round(x.clamp(0, 1) * 255).to(torch.uint8)
    assert got.dtype == torch.uint8 and torch.equal(got, want)


@pytest.mark.parametrize("grid", [16, 32, 128])
def test_coarse_pretest_other_grid_sizes(decoder, scene, grid, monkeypatch):
    """The pre-test adapts to the grid (block = 4 cells; it switches itself off where the coarse table would not fit): random sparse
    bitfields at 16^3 / 32^3 / 128^3, every output bit-identical with the pre-test on and off."""
    from ssdnerf_amd.decoders import pack_triplanes
    g = torch.Generator().manual_seed(grid)
    code = scene["code"].cuda()[None]
    planes = pack_triplanes(code)
    occ = (torch.rand(grid ** 3, generator=g) < 0.02)
    occ.view(grid, grid, grid)[grid // 4: grid // 2, grid // 4: grid // 2, grid // 4: grid // 2] |= torch.rand(grid // 4, grid // 4, grid // 4, generator=g) < 0.5
    bits = torch.from_numpy(np.packbits(occ.numpy().astype(np.uint8), bitorder="little")).cuda()[None]        # morton-ordered index space: any pattern is a valid field
    ro, rd = _view(90)
    o, d = torch.from_numpy(ro).cuda()[None], torch.from_numpy(rd).cuda()[None]

    def run():
        out = decoder.render_packed(planes, o, d, bits, [grid], [0.0], 1e-4, bg_color=1.0, want_counts=True, check_overflow=False)
        return out["image"][0].clone(), out["depth"][0].clone(), decoder.last_render_stats["sample_counts"][0].clone()
    a = run()
    monkeypatch.setenv("SSDNERF_NO_COARSE", "1")
    b = run()
    monkeypatch.delenv("SSDNERF_NO_COARSE")
    assert int((b[2] > 0).sum()) > 100 and int((b[2] == 0).sum()) > 100
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])


@pytest.mark.parametrize("ns,plane_hw", [([70001, 40320], (128, 128)), ([200000, 0, 130001], (128, 128)), ([30011, 50000], (40, 72)), ([66000, 3000, 64000], (128, 128))],
                         ids=["two-scenes", "split-sample-ranges-and-an-empty-scene", "ragged-tiles-non-square", "same-texel-collisions"])
def test_fused_decode_backward_matches_autograd_through_the_eager_decode(ns, plane_hw):
    """d loss / d code of ``point_decode`` with the decoder frozen: fused kernels (ssdnerf_point_decode + _backward: per-sample feature gradient,
    binned LDS reduction over 32 x 32-texel tiles, NCHW sum) vs PyTorch autograd through grid_sample + nn.Linear on the same GPU.  Ragged
    point lists, points outside the box (border clamp), zero upstream gradients; sample ranges split over several blocks per tile (> 64 k
    samples per scene), a scene without samples, planes whose sides are not multiples of the tile."""
    import ssdnerf_amd  # noqa: F401
    from ssdnerf_amd.registry import MODULES
    from ssdnerf_amd import synthetic as S
    dec = MODULES.build(dict(type="TriPlaneDecoder", interp_mode="bilinear", base_layers=[18, 64], density_layers=[64, 1], color_layers=[64, 3],
                             use_dir_enc=True, dir_layers=[16, 64], activation="silu", sigma_activation="trunc_exp", sigmoid_saturation=0.001,
                             max_steps=256))
    dec.load_state_dict(S.make_decoder_params(), strict=False)
    dec = dec.cuda().train(True).requires_grad_(False)
    g = torch.Generator().manual_seed(17)
    if plane_hw == (128, 128):
        code = torch.stack([S.make_triplane(3 + i) for i in range(len(ns))]).cuda()
    else:
        code = (torch.randn(len(ns), 3, 6, *plane_hw, generator=g) * 0.5).cuda()
    xyzs = [(torch.rand(n, 3, generator=g) * 2.3 - 1.15).cuda() for n in ns]
    if ns == [66000, 3000, 64000]:
        # r06 (the reduction's in-wave elections and run sums): 1000 points repeated 66 times in a row (runs that straddle 16-lane rows and waves), every sample of a
        # scene on ONE point, and axis-parallel rays whose samples share a texel on one plane and walk across quadrant and tile borders on the other two
        xyzs[0] = xyzs[0][:1000].repeat_interleave(66, dim=0)
        xyzs[1] = xyzs[1][:1].expand(3000, 3).contiguous()
        o = (torch.rand(500, 1, 3, generator=g) * 2 - 1).cuda()
        t = torch.linspace(-1.0, 1.0, 128).cuda()[None, :, None] * torch.tensor([0.0, 0.0, 1.0]).cuda()
        xyzs[2] = (o * torch.tensor([1.0, 1.0, 0.0]).cuda() + t).reshape(-1, 3).contiguous()
    dirs = [torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).cuda() for n in ns]
    gs = (torch.randn(sum(ns), generator=g) * 0.1).cuda()
    gc = torch.randn(sum(ns), 3, generator=g).cuda()
    gs[1000:3000] = 0; gc[1000:3000] = 0                                   # skipped points

    def run(fused):
        dec.fused_code_grad = fused
        c = code.clone().requires_grad_(True)
        sig, rgb, num = dec.point_decode(xyzs, dirs, c)
        (gcode,) = torch.autograd.grad((sig * gs).sum() + (rgb * gc).sum(), c)
        return sig.detach(), rgb.detach(), num, gcode

    try:
        s1, r1, n1, g1 = run(True)
        s0, r0, n0, g0 = run(False)
        dec.decode_bwd_feat_mfma = True                                        # r06, opt-in: the feature gradients on the matrix cores (bf16-pair products)
        _, _, _, g2 = run(True)
    finally:
        dec.fused_code_grad = True
        dec.decode_bwd_feat_mfma = False
    assert n1 == n0 == ns
    assert float(((s1 - s0).abs() / s0.abs().clamp(min=1e-3)).max()) <= 2e-5 and float((r1 - r0).abs().max()) <= 2e-6
    scale = float(g0.abs().max())
    assert scale > 0 and float((g1 - g0).abs().max()) <= 2e-4 * scale, (float((g1 - g0).abs().max()), scale)
    assert float((g2 - g0).abs().max()) <= 2e-4 * scale and float((g2 - g1).abs().max()) > 0, (float((g2 - g0).abs().max()), scale)      # (same bound; another kernel did run)
    # decoder parameters that need a gradient keep the autograd path
    dec.requires_grad_(True)
    c = code.clone().requires_grad_(True)
    sig, rgb, _ = dec.point_decode(xyzs, dirs, c)
    gw = torch.autograd.grad(sig.sum() + rgb.sum(), dec.base_net[0].weight)[0]
    assert bool(torch.isfinite(gw).all()) and float(gw.abs().max()) > 0


@pytest.mark.parametrize("view,dt_gamma", [(64, 0.0), (200, 0.0038095)])
def test_direction_term_product_count(scene, decoder, view, dt_gamma):
    """The MFMA shading kernel forms all six split products of the direction term by default (SSDNERF_SHADE_FULL_DIR_PRODUCTS) and three of them
    (16 significand bits per factor) on request (TriPlaneDecoder.shade_dir_products = 3).  Both settings against the oracle with the
    SAME tolerance; what does not depend on the direction term -- sample counts, depth, opacity -- bit for bit equal; and how far apart the two
    images are (the bound quoted in include/ssdnerf_hip.h and DESIGN.md)."""
    from oracle import render as R
    ro, rd = _view(view)
    rgb0, dep0, ws0 = R.render_eval(scene["params"], scene["code"], scene["bits"], ro, rd, dt_gamma=dt_gamma)
    res = {}
    for n in (3, 6):
        decoder.shade_dir_products = n
        try:
            res[n] = _render_gpu(decoder, scene, ro, rd, "fused", dt_gamma)
        finally:
            decoder.shade_dir_products = type(decoder).shade_dir_products
        np.testing.assert_allclose(res[n][0], rgb0, rtol=0, atol=2e-5)
        np.testing.assert_allclose(res[n][1], dep0, rtol=0, atol=1e-4)
    assert np.array_equal(res[3][3], res[6][3])
    assert np.array_equal(res[3][1].view(np.uint32), res[6][1].view(np.uint32)) and np.array_equal(res[3][2].view(np.uint32), res[6][2].view(np.uint32))
    diff = np.abs(res[3][0] - res[6][0])
    assert diff.max() > 0.0, "the two settings must be different kernels"
    assert diff.max() <= 5e-6 and diff.mean() <= 2e-7, (diff.max(), diff.mean())
    # the full-precision form is the closer one to the oracle
    assert np.abs(res[6][0] - rgb0).mean() <= np.abs(res[3][0] - rgb0).mean() + 1e-9


def test_specialised_shading_kernels_are_bit_identical(tmp_path):
    """k_shade_mfma has three forms: generic (any grid / plane size), the hot-path geometry as compile-time constants (64^3 grid, 128 x 128
    planes, bound 1, 256 steps), and that with dt_gamma == 0 (constant march step).  The specialised forms only turn operands into literals:
    every output must be bit-identical to the generic form's (SSDNERF_SHADE_GENERIC=1; read once per process, so one interpreter per form)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for name, extra in {"specialised": {}, "generic": {"SSDNERF_SHADE_GENERIC": "1"}}.items():
        path = str(tmp_path / f"{name}.npz")
        env = {k: v for k, v in os.environ.items() if k != "SSDNERF_SHADE_GENERIC"}
        env.update(extra, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        subprocess.run([sys.executable, os.path.join(root, "tests", "_render_variant.py"), path], check=True, env=env, cwd=root, timeout=300)
        outs[name] = np.load(path)
    assert int(outs["generic"]["zero_counts"].sum()) > 100000 and int(outs["generic"]["mixed_counts"].sum()) > 100000
    for k in outs["generic"].files:
        assert np.array_equal(outs["generic"][k], outs["specialised"][k]), k


def test_camera_fed_render_is_bit_identical_to_ray_arrays(decoder, scene):
    """ssdnerf_render_*_cams generate every ray in the kernels (pose + intrinsics -> o, d in registers) with the arithmetic of ssdnerf_cam_rays:
    all outputs, per-ray sample counts included, equal the render of the materialised ray arrays bit for bit (two scenes, different cone
    angles, non-square views whose pixel count is not a power of two, and the 128 x 128 hot-path shape)."""
    from ssdnerf_amd import nerf, synthetic as S
    from ssdnerf_amd.decoders import pack_triplanes
    code = torch.stack([scene["code"], S.make_triplane(12, "uniform")]).cuda()
    bits = torch.from_numpy(scene["bits"]).cuda()[None].expand(2, -1).contiguous()
    planes = pack_triplanes(code, decoder.plane_dtype)
    for (h, w, views) in ((128, 128, [3, 64, 180]), (40, 56, [10, 200])):
        poses = S.spiral_poses()[views].cuda()[None].expand(2, -1, -1, -1).contiguous()
        intr = S.cars_intrinsics(w, h).cuda()[None, None].expand(2, len(views), -1).contiguous()
        ro, rd = nerf.get_cam_rays(poses, intr, h, w)
        a = decoder.render_packed(planes, ro.reshape(2, -1, 3), rd.reshape(2, -1, 3), bits, 64, [0.0, 0.0038095], 1e-4, bg_color=1.0, want_counts=True,
                                  check_overflow=False)
        ca = decoder.last_render_stats["sample_counts"].clone()
        b = decoder.render_packed(planes, None, None, bits, 64, [0.0, 0.0038095], 1e-4, bg_color=1.0, want_counts=True, check_overflow=False,
                                  cams=(poses, intr, h, w))
        cb = decoder.last_render_stats["sample_counts"]
        assert int(ca.sum()) > 10000 and torch.equal(ca, cb)
        for k in ("image", "depth", "weights_sum"):
            assert torch.equal(a[k], b[k]), (h, w, k)


@pytest.mark.parametrize("dt_gamma", [0.0, "tensor"])
def test_batched_train_march_equals_the_per_scene_path(scene, dt_gamma):
    """Train branch of VolumeRenderer.forward: all scenes marched by ssdnerf_march_rays_train_batch_count/_write (one host read, exact-size packed
    arrays, global ray records) vs one ssdnerf_march_rays_train per scene as the reference calls it.  Same samples in the same per-ray order, so the
    composited outputs are bit-identical; the code gradient agrees to rounding (the decode backward's additions are unordered).  Three scenes, one
    of them EMPTY (no occupied cell), a ray count that is not a multiple of the scan block, injected jitter."""
    from ssdnerf_amd.decoders import TriPlaneDecoder
    from ssdnerf_amd import synthetic as S
    dec = TriPlaneDecoder(base_layers=[18, 64], density_layers=[64, 1], color_layers=[64, 3], dir_layers=[16, 64], max_steps=256)
    dec.load_state_dict(scene["params"], strict=False)
    dec = dec.cuda().train(True).requires_grad_(False)
    g = torch.Generator().manual_seed(5)
    n = 3001
    ro, rd = _view(40)
    pick = torch.randperm(ro.shape[0], generator=g)[:n]
    rays_o = torch.from_numpy(ro)[pick][None].repeat(3, 1, 1).cuda()
    rays_d = torch.from_numpy(rd)[pick][None].repeat(3, 1, 1).cuda()
    code = torch.stack([scene["code"], S.make_triplane(5), scene["code"]]).cuda()
    bits = torch.from_numpy(scene["bits"]).cuda()[None].repeat(3, 1)
    bits[1] = 0                                                        # a scene without a single sample
    noises = torch.rand(3, n, generator=g).cuda()
    dtg = torch.tensor([0.0, 0.004, 0.002]).cuda() if dt_gamma == "tensor" else dt_gamma
    target = torch.rand(3, n, 3, generator=g).cuda()

    def run(batched):
        dec.batched_train_march = batched
        dec.injected_noises = noises
        c = code.clone().requires_grad_(True)
        try:
            out = dec(rays_o, rays_d, c, bits, 64, dt_gamma=dtg, perturb=True)
        finally:
            dec.injected_noises = None
        img = out["image"] if isinstance(out["image"], torch.Tensor) else torch.stack(list(out["image"]))
        ws = out["weights_sum"] if isinstance(out["weights_sum"], torch.Tensor) else torch.stack(list(out["weights_sum"]))
        (gcode,) = torch.autograd.grad(((img - target) ** 2).mean() + ws.mean(), c)
        return img.detach().reshape(3, n, 3), ws.detach().reshape(3, n), gcode

    try:
        i1, w1, g1 = run(True)
        i0, w0, g0 = run(False)
    finally:
        dec.batched_train_march = True
    assert float(w0[0].max()) > 0.5 and float(w0[1].abs().max()) == 0.0
    assert torch.equal(i1, i0) and torch.equal(w1, w0)
    scale = float(g0.abs().max())
    assert scale > 0 and float((g1 - g0).abs().max()) <= 1e-5 * scale, (float((g1 - g0).abs().max()), scale)
    assert float(g1[1].abs().max()) == 0.0


def test_view_level_cull_edge_cases(decoder, scene):
    """The view-level cull (k_view_masks + tile test in k_ray_cull) only ever removes rays that the exact path would finish with the background,
    so a camera-fed render must equal the ray-array render (no view cull there) bit for bit for: a camera INSIDE the box and one so close that set
    blocks lie behind its image plane (mask gives up), a zoomed-in view (object larger than the image), an off-centre principal point, a pose
    with a scaled rotation (not rigid: gives up), view sizes that are not multiples of 8 or of 16, and views smaller than 16 pixels (cull off)."""
    from ssdnerf_amd import nerf, synthetic as S
    from ssdnerf_amd.decoders import pack_triplanes
    code = scene["code"].cuda()[None]
    bits = torch.from_numpy(scene["bits"]).cuda()[None]
    planes = pack_triplanes(code, decoder.plane_dtype)
    base = S.spiral_poses()[[5, 77, 140, 222]].clone()
    inside = base[0].clone(); inside[:3, 3] = torch.tensor([0.15, -0.1, 0.2])
    close = base[1].clone(); close[:3, 3] = close[:3, 3] * (1.02 / close[:3, 3].norm())
    scaled = base[2].clone(); scaled[:3, :3] = scaled[:3, :3] * 1.3
    poses = torch.stack([base[0], inside, close, scaled, base[3]]).cuda()[None].contiguous()
    hit_any = 0
    for (h, w, fscale, shift) in ((128, 128, 1.0, 0.0), (64, 64, 3.0, 0.0), (36, 50, 1.0, 7.5), (40, 72, 0.6, -11.0), (12, 12, 1.0, 0.0)):
        intr = S.cars_intrinsics(w, h).clone()
        intr[:2] *= fscale; intr[2] += shift
        intr = intr.cuda()[None, None].expand(1, poses.size(1), -1).contiguous()
        ro, rd = nerf.get_cam_rays(poses, intr, h, w)
        a = decoder.render_packed(planes, ro.reshape(1, -1, 3), rd.reshape(1, -1, 3), bits, 64, [0.0], 1e-4, bg_color=1.0, want_counts=True, check_overflow=False)
        ca = decoder.last_render_stats["sample_counts"].clone()
        b = decoder.render_packed(planes, None, None, bits, 64, [0.0], 1e-4, bg_color=1.0, want_counts=True, check_overflow=False, cams=(poses, intr, h, w))
        cb = decoder.last_render_stats["sample_counts"]
        hit_any += int((ca > 0).sum())
        assert torch.equal(ca, cb), (h, w)
        for k in ("image", "depth", "weights_sum"):
            assert torch.equal(a[k], b[k]), (h, w, k)
    assert hit_any > 5000


def test_render_kernels_write_the_quantised_image_too(decoder, scene):
    """``want_u8``: the camera-fed render kernels store the uint8 image next to the float one; it must equal ssdnerf_quantize_u8 of the float image
    (clamp, x 255, round half to even) for every pixel -- background pixels of the cull / march kernels and shaded pixels alike."""
    from ssdnerf_amd import nerf, synthetic as S
    from ssdnerf_amd.decoders import pack_triplanes
    planes = pack_triplanes(scene["code"].cuda()[None], decoder.plane_dtype)
    bits = torch.from_numpy(scene["bits"]).cuda()[None]
    poses = S.spiral_poses()[[7, 99, 201]].cuda()[None].contiguous()
    intr = S.cars_intrinsics(128, 128).cuda()[None, None].expand(1, 3, -1).contiguous()
    for bg in (1.0, 0.3):
        out = decoder.render_packed(planes, None, None, bits, 64, [0.0], 1e-4, bg_color=bg, check_overflow=False, cams=(poses, intr, 128, 128), want_u8=True)
        assert out["image_u8"].dtype == torch.uint8 and out["image_u8"].shape == out["image"].shape
        assert torch.equal(out["image_u8"], nerf.quantize_u8(out["image"]))
        assert int((out["weights_sum"] > 0.5).sum()) > 1000
