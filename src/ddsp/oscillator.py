from diffsound_amd.ddsp.oscillator import (DampedOscillator, DirectValue, FilteredNoise, GTDampedOscillator,  # noqa: F401
                                           TraditionalDampedOscillator, WeightedParam, WeightedSum, init_damps,
                                           oscillator_bank)
