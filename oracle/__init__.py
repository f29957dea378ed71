"""CPU oracle for the neural-audio-fp hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy, with a float64 and a float32 mode) of
the arithmetic the reference `mimbres/neural-audio-fp` runs on its hot path:

    1-s 8 kHz segment -> STFT -> mel -> log -> 8 x (conv1x3, conv3x1) encoder
    -> divide-and-encode -> L2-normalised 128-d fingerprint, and the NT-Xent
    in-batch contrastive loss.

Every function cites the reference file:line it follows (paths relative to the
reference checkout).  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import anything from here, and only as the
CHECKER.  The product path (`neural-audio-fp_amd/`) never imports this package
and has no CPU fallback.

PARITY UNPINNED.  The reference executes every line of this arithmetic inside
third-party packages that are absent from the reference tree and from this
image: tensorflow (pins disagree: requirements.txt:1 `~=2.2.0`,
environment.yml:138 `2.4.1`), kapre==0.3.5 (requirements.txt:3) and
librosa 0.8.1 (environment.yml:77).  The reference ships no test suite, no
golden vectors and no fixtures for this path, so there is
nothing from the reference itself to pin the oracle against beyond:
  * the parameter-count known answer `Total params: 19,224,576`
    (model/fp/nnfp.py:271), reproduced exactly by `nnfp.count_params((256,63,1))`;
  * the hard-coded input shape (256,32,1) for a 1-s segment (model/fp/nnfp.py:248);
  * 59 segments per 30-s clip implied by eval/test_ids_icassp2021.npy.
What stands in: each stage here is checked in tests/ against an INDEPENDENT
formulation (torch.stft / scipy.fft, torch conv2d with explicit padding,
torch layer_norm, torch cross_entropy, autograd vs analytic gradients), and in
float64 vs float32 to bound rounding; the front end additionally against a
third-party implementation this repository did not write
(transformers.audio_utils.mel_filter_bank / window_function / spectrogram, which
reproduce librosa's Slaney filterbank and an STFT: same 941 taps, |d log-mel| < 1e-6,
tests/test_oracle_melspec.py).  The published algorithms restated from the pinned
third-party versions are named in each docstring.
"""
from . import melspec, nnfp, ntxent, segments, optim, specaug  # noqa: F401
