"""diffsound_amd - MI355X-native (gfx950) modal-sound hot path behind DiffSound's operator API.

Hot stages (hand-written HIP in ``csrc/`` behind the C ABI of ``include/diffsound_hip.h``):
tet-FEM K/M assembly -> block eigensolver (BSR-3 SpMM + MFMA Gram) -> damped-oscillator bank,
forward and backward.  There is no CPU fallback: operations raise if the HIP extension or a HIP
device is missing.
"""
__version__ = "0.1.0"
