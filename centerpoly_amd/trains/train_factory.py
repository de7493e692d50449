"""train_factory (reference: src/lib/trains/train_factory.py:12-18); polydet only."""
from .polydet import PolydetTrainer

train_factory = {"polydet": PolydetTrainer}
