"""Host crops -> embeddings through the overlapped copy / forward pipeline (face_recognition_amd/streaming.py):
same bits as forward_u8 on the same crops, whatever the buffer a batch travelled in."""
import pytest
import torch

from face_recognition_amd.streaming import EmbeddingStream

pytestmark = pytest.mark.gpu


def _crops(n, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 256, (n, 112, 112, 3), generator=g, dtype=torch.uint8)


@pytest.mark.parametrize("tta", [False, True])
def test_stream_equals_forward_u8(hip_model, device, tta):
    sizes = [64, 64, 17, 64, 3, 40, 1]
    batches = [_crops(n, 100 + i) for i, n in enumerate(sizes)]
    want = [hip_model.forward_u8(b.to(device), tta=tta).cpu() for b in batches]
    stream = EmbeddingStream(hip_model, max_batch=64, depth=2, tta=tta)
    got = [e.cpu() for e in stream.embed(batches)]                          # pageable tensors: staged through the stream's buffers
    assert len(got) == len(want) and all(torch.equal(a, b) for a, b in zip(got, want))
    got = [e.cpu() for e in stream.embed([b.pin_memory() for b in batches])]   # pinned tensors: copied from where they are
    assert all(torch.equal(a, b) for a, b in zip(got, want))
    got = [e.cpu() for e in stream.embed(b.numpy() for b in batches)]       # numpy arrays from a decoder
    assert all(torch.equal(a, b) for a, b in zip(got, want))
    # a producer that fills the pinned buffers itself, three buffers deep, results read late
    stream3 = EmbeddingStream(hip_model, max_batch=64, depth=3, tta=tta)
    outs = []
    for i, b in enumerate(batches):
        stream3.host_buffer(i % 3)[:b.shape[0]].copy_(b)
        outs.append(stream3.submit(i % 3, b.shape[0]))
    torch.cuda.synchronize()
    assert all(torch.equal(a.cpu(), b) for a, b in zip(outs, want))


def test_stream_argument_errors(hip_model, device):
    with pytest.raises(ValueError):
        EmbeddingStream(hip_model, depth=1)
    stream = EmbeddingStream(hip_model, max_batch=8)
    with pytest.raises(ValueError):
        stream.submit(0, 9)
    with pytest.raises(RuntimeError):
        stream.submit(0, 4, src=_crops(4, 1))                               # not pinned
    with pytest.raises(RuntimeError):
        stream.submit(0, 4, src=torch.zeros((4, 3, 112, 112), dtype=torch.uint8).pin_memory())
    assert list(stream.embed([])) == []
