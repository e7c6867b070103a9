from diffsound_amd.utils.utils import LOBPCG_solver_freq  # noqa: F401
