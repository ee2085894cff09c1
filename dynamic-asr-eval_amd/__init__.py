"""dynamic-asr-eval on MI355X: the dynamic-eval inner loop of robflynnyh/dynamic-asr-eval
(reference lcasr/lib.py:450-640) rebuilt on hand-written HIP kernels for gfx950 behind the reference's own
Python call convention.  See DESIGN.md for the path, INTEGRATION.md for the drop-in boundary."""
__version__ = "0.1.0"
