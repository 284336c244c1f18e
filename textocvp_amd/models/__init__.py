"""
Host-side mirror of the reference's ``src/models`` operator API: same class
names, constructor kwargs (= JSON config keys), ``forward`` signatures, output dict keys and
``state_dict`` layout, so reference checkpoints load unchanged -- but every forward runs on the
hand-written HIP kernels of libtocvp.so (textocvp_amd.kernels).  Inference only (no autograd).
"""
