from diffsound_amd.ddsp.mss_loss import MSSLoss, SSSLoss  # noqa: F401
