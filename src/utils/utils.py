from diffsound_amd.utils.utils import LOBPCG_solver_freq, plot_spec, resample  # noqa: F401
