"""The parts of bench.py (the driver's contract stays `python bench.py`): common, stream, counters, side, parity."""
