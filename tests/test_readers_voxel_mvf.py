"""CPU: the torch-CPU statement of the voxel and multi-view readers (oracle/torch_readers.py, test infrastructure) against fixtures made by RUNNING
the reference (oracle/gen_golden.py voxel / mvf: det3d/models/readers/voxel_encoder.py and mvf_encoder.py imported unmodified; torch_scatter restated,
spconv stood in for at import only) -- this pins the statement the HIP kernels of csrc/group.hip are compared with on the GPU
(tests/test_gpu_readers_voxel_mvf.py, which also checks the kernels against these fixtures directly).  Indices bit-exact, features within 1e-5
(scatter_mean's summation order).  Plus the host logic of the product modules: state-dict keys, YAML constructor, and that they refuse CPU tensors.
What needs spconv -- SingleView.forward and MVFFeatureNet.forward end to end -- has no fixture (unpinned, like the backbone)."""
import numpy as np
import pytest

from conftest import load_golden

torch = pytest.importorskip("torch")


def test_voxel_statement_matches_the_reference():
    from oracle import torch_readers as R

    g = load_golden("voxel_b3_gap")
    rows, c, inv, grid, mean = R.voxel_net(torch.from_numpy(g["points"]), list(g["voxel_size"]), list(g["pc_range"]))
    assert np.array_equal(c.numpy(), g["coords"]) and c.dtype == torch.int32          # [b, z, y, x]
    assert np.array_equal(inv.numpy(), g["unq_inv"])
    assert np.array_equal(np.asarray(grid), g["grid"])                                  # [gz, gy, gx]
    np.testing.assert_allclose(mean.numpy(), g["features"], rtol=0, atol=1e-5)
    assert set(np.unique(g["coords"][:, 0])) == {0, 2}                                  # the fixture has an empty middle sample


@pytest.mark.parametrize("tag", ["pillar", "cyl"])
def test_mvf_grouping_statements_match_the_reference(tag):
    from oracle import torch_readers as R

    g = load_golden("mvf_parts")
    pts = torch.from_numpy(g["points"])
    if tag == "pillar":
        f, c, inv, grid = R.clamp_view(pts, list(g["voxel_size"]), list(g["pc_range"]))
    else:
        f, c, inv, grid = R.clamp_view(R.cylinder_rows(pts), list(g["cylinder_size"]), list(g["cylinder_range"]))
    assert np.array_equal(c.numpy(), g[f"{tag}_coords"]) and np.array_equal(inv.numpy(), g[f"{tag}_unq_inv"])
    assert np.array_equal(np.asarray(grid), g[f"{tag}_grid"])
    ref = g[f"{tag}_features"]
    assert f.shape == ref.shape == (len(g["points"]), 10)
    assert np.array_equal(f.numpy()[:, :5], ref[:, :5]) and np.array_equal(f.numpy()[:, 8:], ref[:, 8:])     # raw columns and centre offsets: exact
    np.testing.assert_allclose(f.numpy()[:, 5:8], ref[:, 5:8], rtol=0, atol=2e-5)                              # cluster offsets: sum order
    assert len(inv) == len(g["points"])                                                                       # these nets clamp, they drop nothing


def test_pointnet_and_bilinear_match_the_reference():
    from det3d.models.readers.mvf_encoder import PointNet, SingleView
    from oracle import torch_readers as R

    g = load_golden("mvf_parts")
    pn = PointNet(20, 32).eval()
    pn.load_state_dict({k[3:]: torch.from_numpy(np.asarray(v)) for k, v in g.items() if k.startswith("pn_") and k not in ("pn_in", "pn_out")})
    with torch.no_grad():
        np.testing.assert_allclose(pn(torch.from_numpy(g["pn_in"])).numpy(), g["pn_out"], rtol=1e-6, atol=1e-6)
    img, co = torch.from_numpy(g["bil_image"]), torch.from_numpy(g["bil_coords"])
    np.testing.assert_allclose(R.bilinear(img, co).numpy(), g["bil_out"], rtol=1e-6, atol=1e-6)
    # the training-path statement of the product (torch ops over flat indices, differentiable) is the same function
    np.testing.assert_allclose(SingleView.bilinear_interpolate(img, co).numpy(), g["bil_out"], rtol=1e-6, atol=1e-6)


def test_readers_refuse_cpu_tensors():
    """The product modules run on the HIP kernels only: a CPU tensor is an error, not a fallback."""
    from det3d.models.readers.mvf_encoder import CylinderNet, PillarVoxelNet
    from det3d.models.readers.voxel_encoder import VoxelFeatureNet
    from pillarnext_amd._lib import PnxError

    g = load_golden("mvf_parts")
    pts = torch.from_numpy(g["points"])
    for net in (VoxelFeatureNet([0.2, 0.2, 0.4], [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]), PillarVoxelNet(list(g["voxel_size"]), list(g["pc_range"])),
                CylinderNet(list(g["cylinder_size"]), list(g["cylinder_range"]))):
        with pytest.raises(PnxError):
            net(pts)


def test_mvf_net_has_the_reference_keys_and_yaml_constructor():
    from det3d.models.readers.mvf_encoder import MVFFeatureNet

    m = MVFFeatureNet(in_channels=5, voxel_size=[0.075, 0.075, 20], pc_range=[-76.8, -76.8, -10.0, 76.8, 76.8, 10.0], cylinder_size=[0.140625, 0.2, 107],
                      cylinder_range=[-180, -10.0, 0, 180, 10.0, 107], num_filters=[48, 48], layer_nums=[2, 2, 2, 2], ds_layer_strides=[1, 2, 2, 2],
                      ds_num_filters=[48, 96, 192, 192], kernel_size=[3, 3, 3, 3], out_channels=256)     # configs/models/reader/mvf_encoder.yaml
    keys = set(m.state_dict())
    assert {"pillarview.pfn_layers.0.linear.weight", "cylinderview.pfn_layers.1.norm.running_var", "pillarview.blocks.0.0.conv.weight",
            "cylinderview.blocks.3.2.block1.conv.weight", "pillarview.blocks.1.1.norm2.weight", "pointnet1.linear.weight", "pointnet2.norm.bias"} <= keys
    assert m.pointnet1.linear.weight.shape == (192, 20) and m.pointnet2.linear.weight.shape == (256, 576)
    assert m.pillarview.pfn_layers[0].linear.weight.shape == (24, 20) and m.pillarview.pfn_layers[1].linear.weight.shape == (48, 48)
