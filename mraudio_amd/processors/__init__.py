"""Processor call signatures of the reference (``processors/alpro_processors.py``)."""
