"""detector_factory (reference: src/lib/detectors/detector_factory.py:11-17); polydet only."""
from .polydet import PolydetDetector

detector_factory = {"polydet": PolydetDetector}
