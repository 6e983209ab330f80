"""Verification pair sets of the reference's ``Mobilefacenet/data_set`` package (LFW, CFP-FP, AgeDB-30)."""
from .dataloader import LFW, CFP_FP, AgeDB30, img_loader, test_transform, u8_pair_batches  # noqa: F401
