import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def state_dict_np():
    from face_recognition_amd import synthetic
    return synthetic.make_state_dict(1, 512)


@pytest.fixture(scope="session")
def calibrated_state_dict_np(state_dict_np):
    """The seeded weights with the final BatchNorm1d's running statistics set to what the `linear` layer actually
    outputs on sample crops (measured with the CPU oracle), as they are in a trained network: embeddings then spread
    over the sphere instead of clustering around one direction (synthetic.with_embedding_statistics)."""
    import torch
    from face_recognition_amd import synthetic
    from oracle import mobilefacenet_ref
    left, right, _ = synthetic.pair_crops(60, seed=99, noise=0.45)
    taps = {}
    mobilefacenet_ref.forward(state_dict_np, torch.from_numpy(np.concatenate([left, right])), taps)
    y = taps["pre_norm"].numpy().astype(np.float64)
    g, b, mu, var = (np.asarray(state_dict_np["bn." + k], dtype=np.float64) for k in ("weight", "bias", "running_mean", "running_var"))
    z = (y - b) / g * np.sqrt(var + 1e-5) + mu              # the linear layer's output, from the BatchNorm's
    return synthetic.with_embedding_statistics(state_dict_np, z.mean(0), z.var(0))


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda", 0)


@pytest.fixture(scope="session")
def hip_model(device, state_dict_np):
    import torch
    from face_recognition_amd.face_model import MobileFaceNet
    m = MobileFaceNet(512)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state_dict_np.items()})
    return m.to(device).eval()


@pytest.fixture(scope="session")
def hip_model_calibrated(device, calibrated_state_dict_np):
    import torch
    from face_recognition_amd.face_model import MobileFaceNet
    m = MobileFaceNet(512)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in calibrated_state_dict_np.items()})
    return m.to(device).eval()


@pytest.fixture
def invariant_bits():
    """The mode in which an embedding is the same bits whatever batch it is computed in (fra_set_small_batch_mode(0));
    the default mode (1: shortest small-batch launches) is restored afterwards."""
    from face_recognition_amd import face_model
    face_model.set_small_batch_mode(0)
    try:
        yield
    finally:
        face_model.set_small_batch_mode(1)


def cosine_rows(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.sum(a * b, 1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
