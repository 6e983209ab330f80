"""TEST INFRASTRUCTURE -- CPU restatement of the reference's pair-list readers, used only by tests/.

Follows Mobilefacenet/data_set/dataloader.py line by line where it decides results: LFW (:34-52), CFP_FP (:84-96),
AgeDB30 (:129-141), the four-image item (:54-60) and test_lfw_agedb_cfpfp.py:24-26's transform.  Plain loops, numpy
only.  Pinned by nothing the reference holds (it ships no pair list and no image); the expected values in
tests/test_pair_datasets.py are written out by hand from the format rules above.
"""
import numpy as np


def parse_lfw(text):
    names_l, names_r, folds, flags = [], [], [], []
    pairs = text.splitlines()[1:]                                        # :35 header line dropped
    for i, p in enumerate(pairs):
        p = p.split("\t")
        if len(p) == 3:                                                  # :38-42
            name_l = p[0] + "/" + p[0] + "_" + "{:04}.jpg".format(int(p[1]))
            name_r = p[0] + "/" + p[0] + "_" + "{:04}.jpg".format(int(p[2]))
            flag = 1
        elif len(p) == 4:                                                # :43-47
            name_l = p[0] + "/" + p[0] + "_" + "{:04}.jpg".format(int(p[1]))
            name_r = p[2] + "/" + p[2] + "_" + "{:04}.jpg".format(int(p[3]))
            flag = -1
        else:
            raise ValueError("line %d" % (i + 2))
        names_l.append(name_l)
        names_r.append(name_r)
        folds.append(i // 600)
        flags.append(flag)
    return names_l, names_r, folds, flags


def parse_space_list(text, per_fold):
    """CFP_FP (per_fold 700, :86-96) and AgeDB30 (per_fold 600, :131-141)."""
    names_l, names_r, folds, flags = [], [], [], []
    for i, p in enumerate(text.splitlines()):
        p = p.split(" ")
        names_l.append(p[0])
        names_r.append(p[1])
        folds.append(i // per_fold)
        flags.append(int(p[2]))
    return names_l, names_r, folds, flags


def item(img_l, img_r):
    """[L, flip(L), R, flip(R)] of uint8 HWC images; cv2.flip(img, 1) mirrors along the width (:54-60)."""
    return [img_l, img_l[:, ::-1].copy(), img_r, img_r[:, ::-1].copy()]


def transform(img):
    """ToTensor + Normalize(0.5, 0.5): float32 CHW, ((x / 255) - 0.5) / 0.5 in float32 steps."""
    t = (np.transpose(img, (2, 0, 1)).astype(np.float32) / np.float32(255))
    return (t - np.float32(0.5)) / np.float32(0.5)
