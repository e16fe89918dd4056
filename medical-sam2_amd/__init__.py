"""MI355X-native hot path of Medical-SAM2 (per-slice SAM2 forward) -- package root.

Layout:
  csrc/        hand-written HIP kernels (gfx950) + the C-ABI (`include/msam2_hip.h`) -> libmsam2_hip.so
  _lib.py      ctypes loader for the C-ABI library (fails loudly when the library is missing)
  ops.py       thin Python wrappers: torch tensors -> raw pointers/sizes -> C-ABI
  modeling/    host-side mirror of the reference's module interface (same class names, ctor arguments, state-dict
               keys and forward signatures as sam2_train/modeling/**), every forward routed through ops.py
  weights.py   state-dict contract + deterministic name-keyed initialiser
  synthetic.py seeded synthetic images / volumes / prompts (BASELINE.json configs)
"""
__version__ = "0.1.0"
