import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm GPU (runs on the MI355X box)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("2d_object_detection_amd")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no ROCm device")
    return torch.device("cuda:0")


# The YOLOv3 path has two NMS kernels with identical results: one workgroup per (image, class)
# segment, and -- for launches of 256 images or more -- one workgroup per image, which leaves the
# images it cannot take to the first.  Small test batches would never reach the second, so the
# tests of that path run twice: once with the per-image kernel forced on for every launch size
# (YP_IMG_MIN_IMAGES=1) and once with it off.
_BOTH_PATHS = {
    "test_gpu_parity": ("test_v3_detect", "test_v3_write_results", "test_v3_ties", "test_v3_large",
                        "test_v3_empty", "test_v3_output_capacity", "test_v3_randomised", "test_v3_full_size",
                        "test_gpu_matches_committed", "test_v3_multi_set", "test_v3_threshold_ties",
                        "test_v3_batches_in_flight"),
    "test_u_dense_nms": ("test_",),
    "test_w_rescale": ("test_",),
    "test_x_canaries": ("test_",),
    "test_zz_no_alloc": ("test_",),
    "test_y_emulation": ("test_emulated_detect_matches", "test_emulated_capacity", "test_emulated_write_results"),
}


def pytest_generate_tests(metafunc):
    prefixes = _BOTH_PATHS.get(metafunc.module.__name__.rsplit(".", 1)[-1], ())
    if any(metafunc.function.__name__.startswith(p) for p in prefixes):
        if "nms_path" not in metafunc.fixturenames:
            metafunc.fixturenames.append("nms_path")
        metafunc.parametrize("nms_path", ["image", "segment"], indirect=True)


@pytest.fixture
def nms_path(request, monkeypatch):
    monkeypatch.setenv("YP_IMG_MIN_IMAGES", "1" if request.param == "image" else "1000000000")
    return request.param
