from diffsound_amd.ddsp.mss_loss import MSSLoss, SSSLoss, clip_spec, weighted_l1_loss  # noqa: F401
