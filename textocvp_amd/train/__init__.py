"""
Predictor training step (BASELINE configs[4]; reference
04_train_predictor.py:57-108): tape autograd over the HIP kernels (autograd.py), the differentiable
TextOCVP_CustomTF rollout (predictor.py), the frozen SAVi decoder's image loss and slot gradient
(decoder.py) and the clipped-Adam / data-parallel step (step.py).  Coverage and open items: DESIGN.md
section 8.
"""
