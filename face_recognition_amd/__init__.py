"""face_recognition_amd -- MI355X (gfx950) implementation of the MobileFaceNet embedding forward and
facebank matching of liusheng134/Face_Recognition, behind that project's Python interfaces.

    from face_recognition_amd.face_model import MobileFaceNet, l2_norm
    from face_recognition_amd.facebank import load_facebank, prepare_facebank, match_facebank, topk_search
    from face_recognition_amd.Evaluation import getFeature, evaluation_10_fold
    from face_recognition_amd.streaming import EmbeddingStream      # host crops in, copies overlapped with forwards

Importing the package does not touch the GPU; the HIP library is loaded on first use and its absence
is an error (there is no CPU fallback).
"""
__version__ = "0.1.0"
