"""Roofs of the MI355X and the model keys of bench.py.

Part of bench.py's measurement harness (round 6: bench.py was one 1 100-line file running ten legs; the legs live here, bench.py is the
driver entry).  Nothing in this package imports oracle/: the CPU-baseline leg, the only one that may, stays in bench.py."""
HBM_PEAK_GBS = 8000.0          # MI355X spec (MI355X_MICROARCH.md); measured copy peak 6290
FP32_PEAK_TFLOPS = 157.3       # f32 vector == f32-input MFMA dense peak
F16_PEAK_TFLOPS = 2500.0       # dense f16 / bf16 MFMA peak (MI355X_MICROARCH.md)
MODELS = {'m': 'hyperseg-m', 's': 'hyperseg-s', 'sc': 'hyperseg-s-camvid', 'l': 'hyperseg-l', 'lc': 'hyperseg-l-camvid'}
LABELS = {'m': 'HyperSeg-M / EfficientNet-B1 / 1024x512', 's': 'HyperSeg-S / EfficientNet-B1 / 1536x768',
          'sc': 'HyperSeg-S / EfficientNet-B1 / CamVid 768x576', 'l': 'HyperSeg-L / EfficientNet-B3 / 512x512',
          'lc': 'HyperSeg-L / EfficientNet-B1 / CamVid 1024x768 (six-level v1_0 decoder)'}
