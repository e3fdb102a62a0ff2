"""MI355X-native PGD inner step for psychoacoustic adversarial attacks on Wav2Vec2.

Host side (Python on PyTorch-ROCm) of ``libpaa_hip.so``; mirrors the call surface of the
reference's ``src/core`` and ``src/training_utils/train.py`` for the hot path only
(SURVEY.md §8).  Import as ``paa_amd``.
"""
__version__ = "0.1.0"
