"""Processor call signatures of the reference (``processors/alpro_processors.py``)."""
from .alpro_processors import AlproVideoEvalProcessor_Stamps, AlproVideoTrainProcessor_Stamps  # noqa: F401
from .audio_processors import BeatsAudioProcessor  # noqa: F401
