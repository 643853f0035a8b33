"""GPU half of SURVEY.md section 8(f) rank 4 (mesh extraction, lib/core/utils/nerf_utils.py:64-112): the density volume against the oracle's
decode, the marching-cubes kernels against the host walker and analytic surfaces, ``extract_geometry`` end to end."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _decoder():
    from ssdnerf_amd import synthetic as S
    from ssdnerf_amd.decoders import TriPlaneDecoder
    dec = TriPlaneDecoder(interp_mode="bilinear", base_layers=[18, 64], density_layers=[64, 1], color_layers=[64, 3], use_dir_enc=True, dir_layers=[16, 64],
                          activation="silu", sigma_activation="trunc_exp", sigmoid_saturation=0.001, max_steps=256)
    dec.load_state_dict(S.make_decoder_params(), strict=False)
    return dec.cuda().eval()


def test_density_volume_matches_oracle_at_128():
    """``nerf.extract_density_volume`` (the lattice ``extract_geometry`` marches over: the box grown by 0.1 per side, sigma forced to 0 outside the
    AABB, nerf_utils.py:98-112) at 128^3 against the oracle's point decode (PyTorch-CPU grid_sample + nn.Linear) of the same lattice."""
    from oracle.decoder import point_decode
    from ssdnerf_amd import nerf, synthetic as S
    dec, code = _decoder(), S.make_triplane(2021)
    res = 128
    u = nerf.extract_density_volume(dec, code.cuda(), resolution=res).cpu()
    lin = torch.linspace(-1.1, 1.1, res)
    xx, yy, zz = torch.meshgrid(lin, lin, lin, indexing="ij")
    pts = torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], dim=-1)
    with torch.no_grad():
        sig, _ = point_decode(S.make_decoder_params(), code, pts, None, density_only=True)
    outside = (pts.abs() > 1).any(dim=-1)
    want = sig.masked_fill(outside, 0).reshape(res, res, res)
    assert float(u[0].abs().max()) == 0 and float(u[:, :, -1].abs().max()) == 0          # the margin outside the AABB
    scale = float(want.abs().max())
    assert scale > 10                                                                     # the object is there (sigma ~ 40 inside)
    assert float((u - want).abs().max()) <= 2e-5 * scale
    assert int((u > 10).sum()) == int((want > 10).sum())                                  # the inside set at the mesh threshold


def test_marching_cubes_kernels_equal_the_host_walker():
    """Same vertices (bit for bit: same IEEE interpolation), same triangles in the same order, on a volume of white noise inside a negative shell
    (every ambiguous configuration occurs) and on a non-cubic lattice."""
    from ssdnerf_amd import mesh as M
    rng = np.random.default_rng(3)
    for shape in ((14, 14, 14), (9, 17, 12)):
        vol = np.full(shape, -5, np.float32)
        vol[1:-1, 1:-1, 1:-1] = rng.standard_normal(tuple(s - 2 for s in shape))
        v0, t0 = M.marching_cubes_reference(vol, 0.25)
        v, t = M.marching_cubes(torch.from_numpy(vol).cuda(), 0.25)
        assert np.array_equal(v.cpu().numpy().view(np.uint32), v0.view(np.uint32)) and np.array_equal(t.cpu().numpy(), t0)
    empty_v, empty_t = M.marching_cubes(torch.zeros(8, 8, 8, device="cuda"), 1.0)
    assert empty_v.shape == (0, 3) and empty_t.shape == (0, 3)


def test_marching_cubes_of_a_sphere_at_256():
    """The reference's resolution: 256^3 lattice points, 16.6 M cells, in two launches; closed oriented manifold of genus 0, area and enclosed
    volume of the sphere to discretisation error, every vertex on the iso-surface of the trilinear field (exactly on an edge)."""
    from ssdnerf_amd import mesh as M
    n, r = 256, 90.0
    g = torch.arange(n, dtype=torch.float32, device="cuda")
    X, Y, Z = torch.meshgrid(g, g, g, indexing="ij")
    vol = r - torch.sqrt((X - 127.3) ** 2 + (Y - 128.1) ** 2 + (Z - 126.6) ** 2)
    v, t = M.marching_cubes(vol, 0.0)
    st = M.mesh_stats(v.cpu().numpy(), t.cpu().numpy())
    assert st["closed_and_oriented"] and st["euler"] == 2 and st["degenerate"] == 0
    assert abs(st["area"] / (4 * np.pi * r * r) - 1) < 2e-3 and abs(st["volume"] / (4 / 3 * np.pi * r ** 3) - 1) < 2e-3
    rad = torch.sqrt((v[:, 0] - 127.3) ** 2 + (v[:, 1] - 128.1) ** 2 + (v[:, 2] - 126.6) ** 2)
    assert float((rad - r).abs().max()) < 2e-3                                           # linear interpolation of a distance field along an edge


def test_extract_geometry_end_to_end():
    """codes -> density volume (fused decode) -> marching cubes (GPU) -> world coordinates: the synthetic car-sized box comes out as closed
    surfaces inside the AABB, with every vertex at the threshold density."""
    from ssdnerf_amd import mesh as M, nerf, synthetic as S
    dec, code = _decoder(), S.make_triplane(2021).cuda()
    verts, tris = nerf.extract_geometry(dec, code, resolution=128, threshold=10)
    assert verts.dtype == np.float64 and tris.shape[1] == 3 and len(tris) > 1000
    assert np.abs(verts).max() <= 1.0 + 1e-6                                             # sigma is 0 outside the AABB, so nothing crosses 10 out there
    st = M.mesh_stats(verts, tris)
    assert st["closed_and_oriented"] and st["volume"] > 0
    # the reference's index -> world map: v / (res - 1) * (b_max - b_min) + b_min with the box grown by 0.1
    u = nerf.extract_density_volume(dec, code, resolution=128)
    v_idx, _ = M.marching_cubes(u, 10.0)
    np.testing.assert_allclose(verts, v_idx.cpu().numpy().astype(np.float64) / 127.0 * 2.2 - 1.1, rtol=0, atol=1e-6)   # (the box corners are fp32: -1.1f)
    # a vertex lies on one lattice edge; the linear interpolant of the two corner densities there is the threshold
    vi = v_idx.cpu().numpy()
    lo = np.floor(vi).astype(np.int64)
    frac = vi - lo
    axis = frac.argmax(axis=1)
    hi = lo.copy()
    hi[np.arange(len(hi)), axis] += (frac.max(axis=1) > 0)
    uc = u.cpu().numpy()
    a, b = uc[lo[:, 0], lo[:, 1], lo[:, 2]], uc[hi[:, 0], hi[:, 1], hi[:, 2]]
    val = a + (b - a) * frac.max(axis=1)
    assert np.abs(val - 10.0).max() <= 1e-3 * max(1.0, float(np.abs(np.stack([a, b])).max()))
