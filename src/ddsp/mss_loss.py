from diffsound_amd.ddsp.mss_loss import (MSSLoss, SSSLoss, clip_spec, normlize, spec2point,  # noqa: F401
                                         weighted_l1_loss)
