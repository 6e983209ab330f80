"""The counterpart of the reference's test_lfw_agedb_cfpfp.py (load ckpt['net_state_dict'], bs 50, flip,
cos_distance, one line per phase): data layout and checkpoint forms on the CPU, the harness itself on the GPU
against the oracle."""
import io
import re

import numpy as np
import pytest
import torch

from face_recognition_amd import synthetic
from face_recognition_amd import test_lfw_agedb_cfpfp as harness
from face_recognition_amd.Evaluation import getFeature
from face_recognition_amd.face_model import MobileFaceNet


def test_pair_sets_have_the_reference_layout():
    loaders, sizes, sets = harness.load_test_data(50)
    assert sizes == {"LFW": 6000, "CFP_FP": 7000, "AgeDB30": 6000}          # dataloader.py:22-160
    cfp = sets["CFP_FP"]
    assert cfp.folds[0] == 0 and cfp.folds[699] == 0 and cfp.folds[700] == 1 and cfp.folds[6999] == 9   # i // 700 (:90)
    assert sets["LFW"].folds[599] == 0 and sets["LFW"].folds[600] == 1                                  # i // 600 (:41)
    assert set(cfp.flags) == {1, -1} and sum(cfp.flags) == 0
    assert len(loaders["CFP_FP"]) == 140 and len(loaders["LFW"]) == 120                                 # bs 50 (:60)
    small = harness.SyntheticPairs(40, seed=3)
    a, b = small.batch(20, 50), small.batch(20, 50)
    assert len(a) == 4 and a[0].shape == (20, 3, 112, 112) and a[0].dtype == torch.float32
    assert all(torch.equal(x, y) for x, y in zip(a, b))                      # a batch is a function of its arguments
    assert torch.equal(a[1], torch.flip(a[0], dims=[3])) and torch.equal(a[3], torch.flip(a[2], dims=[3]))
    assert float(a[0].abs().max()) <= 1.0
    got = list(harness.PairLoader(small, 16))
    assert [g[0].shape[0] for g in got] == [16, 16, 8]
    # a pair's crops do not depend on the batch size it is served in
    assert torch.equal(torch.cat([g[2] for g in got]), torch.cat([g[2] for g in harness.PairLoader(small, 7)]))


def test_checkpoint_forms(tmp_path):
    """test_lfw_agedb_cfpfp.py:62 reads ckpt['net_state_dict']; DataParallel checkpoints carry 'module.' (:63-67)."""
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(3, 512).items()}
    forms = {"plain": sd, "wrapped": {"iters": 555000, "net_state_dict": sd},
             "module": {"net_state_dict": {"module." + k: v for k, v in sd.items()}}}
    for name, obj in forms.items():
        path = tmp_path / f"{name}.ckpt"
        torch.save(obj, path)
        m = MobileFaceNet(512)
        m.load_state_dict(harness.load_checkpoint(str(path)))
        assert torch.equal(m.state_dict()["conv_4.model.3.conv_dw.bn.running_var"], sd["conv_4.model.3.conv_dw.bn.running_var"])


@pytest.mark.gpu
def test_harness_against_oracle(hip_model_calibrated, device, calibrated_state_dict_np):
    """Three phases of 1000 / 1400 / 1000 pairs (fold = i // 100, 140, 100; sized so that the fp32 oracle embeds the
    13,600 crops in about half a minute) against the fp32 oracle on the same crops:
    accuracy at the oracle's thresholds within 0.1 pt for every phase; the printed line matches what run() returns."""
    from oracle import evaluation_ref, mobilefacenet_ref
    sizes = {"LFW": 1000, "CFP_FP": 1400, "AgeDB30": 1000}
    out = io.StringIO()
    res = harness.run(hip_model_calibrated, device, batch_size=50, sizes=sizes, seed=5, noise=0.45, out=out)
    lines = out.getvalue().strip().splitlines()
    assert len(lines) == 3
    _, _, sets = harness.load_test_data(50, sizes=sizes, seed=5, noise=0.45, device=device)
    embed = lambda x: mobilefacenet_ref.forward(calibrated_state_dict_np, x)      # noqa: E731
    for line, (phase, ds) in zip(lines, sets.items()):
        m = re.match(r"Epoch 1/1，(\w+) average acc:([\d.]+) average threshold:(-?[\d.]+)", line)
        assert m and m.group(1) == phase, line
        cpu_batches = ([t.cpu() for t in b] for b in harness.PairLoader(ds, 200))
        ol, orr = evaluation_ref.get_feature(embed, cpu_batches, flip=True)
        oacc, othr = evaluation_ref.evaluation_10_fold(ol.numpy(), orr.numpy(), ds.folds, ds.flags, "cos_distance")
        # the criterion: accuracy AT THE REFERENCE'S THRESHOLDS within 0.1 pt -- the GPU features scored with the
        # per-fold thresholds the oracle's features select
        fl, fr = getFeature(hip_model_calibrated, harness.PairLoader(ds, 50), device, flip=True)
        at_ref, _ = evaluation_ref.evaluation_10_fold(fl.numpy(), fr.numpy(), ds.folds, ds.flags, "cos_distance",
                                                      fixed_thresholds=othr)
        print(line, "| oracle acc %.4f thr %.4f | GPU features at the oracle's thresholds %.4f" %
              (oacc.mean() * 100, othr.mean(), at_ref.mean() * 100))
        assert abs(at_ref.mean() - oacc.mean()) * 100 <= 0.1 + 1e-9
        # each side's OWN threshold is the mean of a plateau of equally good grid points and can jump when a tie
        # between plateaus breaks, taking test accuracy with it; on these hard, small phases (94-95 %, one pair = 0.1 pt)
        # that is measured at 0.0 - 0.3 pt (three pairs) and bounded at 0.5 here -- the 0.1-pt bar with each side's own
        # thresholds is asserted at LFW's size and difficulty in test_reference_protocol_at_lfw_size_and_difficulty
        assert abs(float(m.group(2)) - oacc.mean() * 100) <= 0.5
        assert abs(float(m.group(3)) - othr.mean()) < 0.05
        assert abs(res[phase][0].mean() * 100 - float(m.group(2))) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("weights", ["calibrated", "uncalibrated"])
def test_reference_protocol_at_lfw_size_and_difficulty(weights, request, device):
    """The reference's protocol as it stands -- thresholds searched on each side, nothing injected -- on a phase of LFW's
    real size (6000 pairs, folds of 600: one pair is 0.017 pt) and LFW's real difficulty for this network (pair noise 0.30:
    the fp32 oracle scores 99.3-99.4 %): the harness's printed accuracy within 0.1 pt of the oracle's.
    Measured: -0.05 pt with calibrated final-BatchNorm statistics, 0.00 pt with the arbitrary ones of the seeded weights
    (whose embeddings cluster; the threshold moves by 0.0005 there).  On HARDER synthetic sets (noise 0.45, the oracle at
    93.4-94.6 %, ten times as many pairs next to the threshold) the same comparison measures -0.12 .. +0.04 pt at 6000 -
    8400 pairs and up to +-0.15 pt at 2000: that figure is recorded, not asserted at 0.1 -- a threshold that is the mean of
    a plateau of equally good grid points moves when one pair's score moves by 0.0006, whatever computed it
    (tests/diag/own_threshold_delta.py prints all of these)."""
    from oracle import evaluation_ref, mobilefacenet_ref
    sd = request.getfixturevalue("calibrated_state_dict_np" if weights == "calibrated" else "state_dict_np")
    model = request.getfixturevalue("hip_model_calibrated" if weights == "calibrated" else "hip_model")
    sizes = {"LFW": 6000}
    out = io.StringIO()
    res = harness.run(model, device, batch_size=50, sizes=sizes, seed=5, noise=0.30, out=out)
    m = re.match(r"Epoch 1/1，LFW average acc:([\d.]+) average threshold:(-?[\d.]+)", out.getvalue().strip())
    assert m, out.getvalue()
    _, _, sets = harness.load_test_data(50, sizes=sizes, seed=5, noise=0.30, device=device)
    ds = sets["LFW"]
    cpu_batches = ([t.cpu() for t in b] for b in harness.PairLoader(ds, 200))
    ol, orr = evaluation_ref.get_feature(lambda x: mobilefacenet_ref.forward(sd, x), cpu_batches, flip=True)
    oacc, othr = evaluation_ref.evaluation_10_fold(ol.numpy(), orr.numpy(), ds.folds, ds.flags, "cos_distance")
    print("%s: harness %s / %s | oracle %.4f / %.4f" % (weights, m.group(1), m.group(2), oacc.mean() * 100, othr.mean()))
    assert oacc.mean() > 0.985                                              # the set is as easy as the real one
    assert abs(float(m.group(1)) - oacc.mean() * 100) <= 0.1 + 1e-9
    assert abs(res["LFW"][0].mean() * 100 - oacc.mean() * 100) <= 0.1 + 1e-9
    assert abs(float(m.group(2)) - othr.mean()) < 0.005


@pytest.mark.gpu
def test_harness_full_size_phases_run(hip_model, device):
    """The reference's sizes -- 6000, 7000 (fold = i // 700) and 6000 pairs at batch size 50 with flip."""
    out = io.StringIO()
    res = harness.run(hip_model, device, batch_size=50, seed=1, out=out)
    assert list(res) == ["LFW", "CFP_FP", "AgeDB30"] and len(out.getvalue().strip().splitlines()) == 3
    for accs, thr in res.values():
        assert accs.shape == (10,) and np.isfinite(accs).all() and np.isfinite(thr).all() and 0.5 < accs.mean() <= 1.0
