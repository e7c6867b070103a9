from diffsound_amd.ddsp.oscillator import (DampedOscillator, DirectValue, TraditionalDampedOscillator,  # noqa: F401
                                           WeightedParam, WeightedSum, init_damps, oscillator_bank)
