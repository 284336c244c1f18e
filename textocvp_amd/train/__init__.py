"""
Predictor training step (BASELINE configs[4]): tape autograd over the HIP
kernels (autograd.py), the differentiable TextOCVP predictor / frozen SAVi decoder (predictor.py) and the
optimiser step (optim.py).  In progress: see DESIGN.md section 8 for what is covered.
"""
