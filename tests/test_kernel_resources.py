"""Every kernel of the library against a committed table of its resource use (vector registers,
scratch bytes per lane, LDS bytes per workgroup, waves per SIMD by registers), taken from the
compiler's own remarks.  A change that makes a kernel spill, grow its LDS block or lose
occupancy shows up here, on the CPU, before the build meets a GPU; a deliberate change updates
the table (python tools/kernel_resources.py > tests/golden/kernel_resources.json)."""
import importlib.util
import json
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLE = os.path.join(ROOT, "tests", "golden", "kernel_resources.json")


def _tool():
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "tools", "kernel_resources.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def built():
    if not os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("c++filt") is None:
        pytest.skip("hipcc / c++filt are not installed here")
    return _tool().collect([])


def test_shipped_kernels_stay_in_their_resource_class(built):
    want = json.load(open(TABLE))
    assert sorted(built) == sorted(want), sorted(set(built) ^ set(want))
    tool = _tool()
    assert tool.check(built, want) == []
    # the table is not allowed to be laxer than the build either (it documents the build)
    for name, got in built.items():
        assert got["scratch"] == want[name]["scratch"] and got["lds"] == want[name]["lds"], name
        assert got["occupancy"] == want[name]["occupancy"], name


def test_hot_kernels_of_the_headline_path_do_not_spill(built):
    """cfg2 / cfg4 run the single-wave front kernel, the per-image NMS kernel and the per-image
    row gather: no scratch, and two per-image NMS workgroups must fit the 160 KB LDS of a CU."""
    for name in ("k_front_heads_v3_wave<5, 3>", "k_image_nms<0, 512, 1024, 2560>", "k_image_nms<0, 512, 1728, 4096>", "k_gather_image", "k_gather_rows"):
        assert built[name]["scratch"] == 0, name
    assert 2 * built["k_image_nms<0, 512, 1728, 4096>"]["lds"] <= 160 * 1024
    assert 2 * built["k_image_nms<0, 512, 1024, 2560>"]["lds"] <= 104 * 1024   # (a third of the LDS stays free)
    assert built["k_image_nms<0, 512, 1024, 2560>"]["vgprs"] <= 80


def test_check_refuses_a_build_that_leaves_the_table():
    tool = _tool()
    want = json.load(open(TABLE))
    worse = json.loads(json.dumps(want))
    worse["k_graph_pairs<0, 32>"]["scratch"] = 132
    worse["k_graph_pairs<0, 32>"]["occupancy"] = 5          # (more waves is not the problem ...)
    worse["k_image_nms<0, 512, 1728, 4096>"]["lds"] += 4096
    worse["k_new_kernel"] = dict(want["k_bbox_iou"])
    bad = tool.check(worse, want)
    assert len(bad) == 3 and any("scratch 132" in b for b in bad) and any("not in the table" in b for b in bad)
