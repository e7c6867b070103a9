"""Import-path compatibility: the reference's experiments do ``from src.diffelastic.diff_model import ...``,
``from src.ddsp.oscillator import ...``, ``from src.lobpcg import ...`` (reference
experiments/material_sync_train.py:14-20).  These modules only re-export ``diffsound_amd``."""
