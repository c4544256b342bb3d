"""Model classes with the reference's names (``models/xinstructblip.py``, ``models/videollama.py``)."""
