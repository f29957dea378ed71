"""MI355X-native hot path of neural audio fingerprinting.

Package directory: `neural-audio-fp_amd/`; import it as `neural_audio_fp_amd`
(the repo-root shim `neural_audio_fp_amd.py` registers this directory under that
name).  Contents:

  csrc/     hand-written HIP kernels for gfx950 + the C ABI (include/nafp.h)
  _lib.py   ctypes binding of libnafp.so (fails loudly when it is not built)
  model/    host-side mirror of the reference's operator interface for this path
            (get_melspec_layer, get_fingerprinter, NTxentLoss, OnlineTripletLoss, LAMB, Dataset /
            genUnbalSequence, trainer, generate_fingerprint)
  eval/     eval_faiss mirror on the exact HIP index (load_memmap_data, get_index, eval_faiss)
"""
from . import _lib  # noqa: F401
from .model.fp.melspec.melspectrogram import Melspec_layer, get_melspec_layer  # noqa: F401
from .model.fp.nnfp import FingerPrinter, get_fingerprinter  # noqa: F401
from .model.fp.NTxent_loss_single_gpu import NTxentLoss  # noqa: F401

__all__ = ['Melspec_layer', 'get_melspec_layer', 'FingerPrinter', 'get_fingerprinter', 'NTxentLoss']
